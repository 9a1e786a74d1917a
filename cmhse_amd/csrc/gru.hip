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

namespace cmhse {

struct GruStepParams {
  const uint64_t* x_rows;
  const uint64_t* tok_rows;
  const float* emb;
  const uint64_t* h0_rows;
  const int32_t* lens;
  const int32_t* out_row;
  const float* w_ih;
  const float* w_hh;
  const float* b_ih;
  const float* b_hh;
  float* hs;
  float* out;
  const float* w_ih_s;  // bf16x3 pre-split weights (rows of split_ld(K) float units) or NULL
  const float* w_hh_s;
  const float* xs;      // bf16x3: pre-split input rows, packed row p at xs + p * split_ld(I)
  float* hs_s;          // bf16x3: pre-split hidden states, packed row p at hs_s + p * split_ld(H)
  const float* h0_s;    // bf16x3: pre-split initial hidden states, sorted sequence s at h0_s + s * split_ld(H)
  float* gates;     // [sumT, 4H] r,z,n,(W_hn h + b_hn) per packed row, or NULL (inference)
  int32_t* argmax;  // [S, H] step of the running maximum (max pooling, training), or NULL
  int32_t I, H, t, S_t, vocab, pool_mode, n_tiles, x_step;
  // mid-size step (gru_step_mid_kernel): hoisted input projection x W_ih^T of the small-batch steps,
  // row (off_cur + m - gx_p0) for an ordinary input, row m (the sorted sequence) for a
  // time-constant one
  const float* gx;
  int64_t gx_p0;
  int32_t gx_per_seq;
  int64_t off_prev, off_cur;
};

// Up to kMaxJobs independent GRU chains share one launch per time step: workgroups
// [start[k], start[k+1]) belong to job k.  Halves (or better) the number of dependent launches and
// of partially filled last waves of workgroups when two encoders run side by side.
constexpr int kMaxJobs = CMHSE_MAX_JOBS;
struct GruStepGroup {
  GruStepParams j[kMaxJobs];
  uint32_t start[kMaxJobs];
  int32_t n;
};

#ifdef TILE_TRACE_BUILD
// Timing-only debug build (tools/tile_trace.py): per-workgroup stamps of the tiled step —
// [0] first instruction, [1] K loops start, [2] after the kernarg reads, [3] K loops end,
// [4] state stores drained (s_memrealtime, 10 ns); [5]/[7] s_memtime at [1]/[3]; [6] HW_ID | XCC_ID << 32.
__device__ uint64_t* g_trace = nullptr;
#define TRACE_MARK(i)                                                              \
  do {                                                                             \
    if (threadIdx.x == 0 && g_trace) g_trace[static_cast<size_t>(blockIdx.x) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define TRACE_MARK(i) do {} while (0)
#endif

__device__ __forceinline__ int group_job(const GruStepGroup& g, unsigned* bx) {
  int ji = 0;
#pragma unroll
  for (int k = 1; k < kMaxJobs; ++k)
    if (k < g.n && blockIdx.x >= g.start[k]) ji = k;
  *bx = blockIdx.x - g.start[ji];
  return ji;
}

// Gate nonlinearities on the hardware exp/rcp units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the
// epilogue evaluates 3 of them per (sequence, unit) per step, and the libm-accurate forms cost
// ~6 % of the step kernel.  Absolute error ~1e-7, far inside the 1e-4 parity bar.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1); saturates cleanly for |x| large (exp -> inf or 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

__device__ __forceinline__ bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

constexpr int kGruBU = 64;   // hidden units per workgroup (x3 gates = 192 weight rows)

// What a tile of the step-CHAIN kernel (gru_step_chain_kernel below) waits for and signals: the
// counter of the same row tile one step earlier must have reached `need` (all its column tiles)
// before the h phase starts, and `done` is bumped once this tile's state rows have left the CU.
struct ChainDep {
  const unsigned* wait;   // NULL: nothing to wait for (the chain's first step)
  unsigned need;
  unsigned* done;
  GridSync sync;          // abort word / status word / timeout of the launch (counter unused)
};

// One tile of the LDS-tiled GRU step: sequences [m0, m0 + BM) x hidden units [u0, u0 + BU) of step
// `t`.  CHAIN = false: the body of gru_step_kernel (one launch per time step).  CHAIN = true: the
// same arithmetic inside gru_step_chain_kernel — the x phase (which does not depend on the
// previous step) first, then the wait for the previous step's rows, the h phase, and the new state
// written THROUGH the non-coherent L2 (agent-scope stores) before `done` is signalled.
template <bool VEC, int MSUB, bool BF3, bool CHAIN>
__device__ __forceinline__ void gru_step_tile(const GruStepParams& p, const unsigned wg, const int t,
                                              const int S_t, const int64_t off_prev,
                                              const int64_t off_cur, const ChainDep& dep) {
  constexpr int BM = 64 * MSUB, BU = kGruBU, BNR = 3 * BU;
#ifdef TILE_TRACE_BUILD
  const uint64_t t_first = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  TRACE_MARK(2);
#ifdef TILE_TRACE_BUILD
  if (threadIdx.x == 0 && g_trace) {
    g_trace[static_cast<size_t>(blockIdx.x) * 8 + 6] =
        static_cast<uint64_t>(__builtin_amdgcn_s_getreg((31 << 11) | 4)) |
        (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32);
    g_trace[static_cast<size_t>(blockIdx.x) * 8 + 0] = t_first;
  }
#endif
  // 1-D grid, N tile fastest: blocks b and b+8 land on the same XCD (round-robin dispatch), so
  // with H/BU a multiple of 8 every XCD's L2 keeps re-serving the same two weight-row slices.
  const int u0 = static_cast<int>(wg % p.n_tiles) * BU;
  const int m0 = static_cast<int>(wg / p.n_tiles) * BM;
  const int srow = tid >> 2;
  const int I = p.I, H = p.H;

  // Rows this thread stages.  A: sequences m0 + srow + 64 i.  B: gate g, unit u0 + (row % BU).
  // Out-of-range rows are clamped to a valid row and flagged invalid (read as zeros).
  rowaddr_t ax[BM / 64];
  rowaddr_t ah[BM / 64];
  bool av[BM / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i) {
    const int m = m0 + srow + 64 * i;
    av[i] = m < S_t;
    const int mc = av[i] ? m : (S_t - 1);
    if (BF3) {
      ax[i] = row_addr(p.xs + (off_cur + mc) * split_ld(I));   // (token lookups included)
    } else if (p.tok_rows != nullptr) {
      long long tok = reinterpret_cast<const long long*>(p.tok_rows[mc])[t];
      tok = tok < 0 ? 0 : (tok >= p.vocab ? p.vocab - 1 : tok);
      ax[i] = row_addr(p.emb + tok * I);
    } else {
      ax[i] = p.x_rows[mc] + static_cast<rowaddr_t>(t) * p.x_step * 4u;
    }
    if (t > 0)
      ah[i] = BF3 ? row_addr(p.hs_s + (off_prev + mc) * split_ld(H))
                  : row_addr(p.hs + (off_prev + mc) * H);
    else if (p.h0_rows != nullptr)
      ah[i] = BF3 ? row_addr(p.h0_s + static_cast<int64_t>(mc) * split_ld(H)) : p.h0_rows[mc];
    else
      ah[i] = row_addr(p.w_hh);  // unused: the h phase is skipped
  }
  rowaddr_t bx[BNR / 64];
  rowaddr_t bh[BNR / 64];
  bool bv[BNR / 64];
#pragma unroll
  for (int i = 0; i < BNR / 64; ++i) {
    const int br = srow + 64 * i;
    const int g = br / BU, u = u0 + (br % BU);
    bv[i] = u < H;
    const int uc = bv[i] ? u : (H - 1);
    if (BF3) {
      bx[i] = row_addr(p.w_ih_s + (static_cast<int64_t>(g) * H + uc) * split_ld(I));
      bh[i] = row_addr(p.w_hh_s + (static_cast<int64_t>(g) * H + uc) * split_ld(H));
    } else {
      bx[i] = row_addr(p.w_ih + (static_cast<int64_t>(g) * H + uc) * I);
      bh[i] = row_addr(p.w_hh + (static_cast<int64_t>(g) * H + uc) * H);
    }
  }

  // accumulators per 32-sequence sub-tile: 0 = r, 1 = z, 2 = W_in x, 3 = W_hn h
  f32x16 acc[MSUB][4];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[ms][a] = zero16();

  const int a_row0 = wm * 32 * MSUB;
  const int b_row0[3] = {0 * BU + wn * 32, 1 * BU + wn * 32, 2 * BU + wn * 32};
  const bool have_h = (t > 0) || (p.h0_rows != nullptr);

  const int pool_mode = p.pool_mode;
  float* const hs = p.hs;
  float* const gates = p.gates;
  float* const out = p.out;
  int32_t* const argmax = p.argmax;
  const uint64_t* const h0_rows = p.h0_rows;
  const int32_t* const out_row = p.out_row;
  const int32_t* const lens = p.lens;
  TRACE_MARK(1);
#ifdef TILE_TRACE_BUILD
  if (threadIdx.x == 0 && g_trace) g_trace[static_cast<size_t>(blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
  if (BF3) {
    // pre-split A operands: xs, then hs_s of the previous step (or the pre-split initial states)
    nt_phase_bf3<BM, BNR, MSUB, 3, 4, 2, true>(smem, ax, av, bx, bv, I, a_row0, b_row0, acc);
    if (have_h) nt_phase_bf3<BM, BNR, MSUB, 3, 4, 3, true>(smem, ah, av, bh, bv, H, a_row0, b_row0, acc);
  } else {
    nt_phase<BM, BNR, MSUB, 3, 4, 2, VEC>(smem, ax, av, bx, bv, I, a_row0, b_row0, acc);
    if (CHAIN) {
      // the previous step's rows of this row tile: complete (written through by their tiles)?
      if (dep.wait != nullptr && !flag_wait(dep.sync, dep.wait, dep.need)) return;
    }
    if (have_h) nt_phase<BM, BNR, MSUB, 3, 4, 3, VEC>(smem, ah, av, bh, bv, H, a_row0, b_row0, acc);
  }
  TRACE_MARK(3);
#ifdef TILE_TRACE_BUILD
  if (threadIdx.x == 0 && g_trace) g_trace[static_cast<size_t>(blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: gates, state update, pooling ----
  // The operands that do not come from the MFMA chain — the previous state of this lane's 16
  // (sequence, unit) elements and the four bias terms — are requested all at once, branch-free
  // (clamped indices): ONE memory round trip per sub-tile instead of one per element.  The gate
  // math is then straight-line with predicated stores.  (Requesting them before the K loops would
  // hide that trip too, but the 20 extra live registers cost the third wave per SIMD.)
  const int u = u0 + wn * 32 + acc_col(lane);
  const bool uv = u < H;
  const int uc = uv ? u : (H - 1);
  const float b_r = p.b_ih[uc] + p.b_hh[uc];
  const float b_z = p.b_ih[H + uc] + p.b_hh[H + uc];
  const float b_in = p.b_ih[2 * H + uc];
  const float b_hn = p.b_hh[2 * H + uc];
  // previous states of BOTH 32-row sub-tiles first: the stores of sub-tile 0
  // may alias the loads of sub-tile 1 as far as the compiler knows, so left inside the loop below
  // the second round trip starts only after the first sub-tile's gate math and stores
  float hp_all[MSUB][16];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) {
    const int mrow0 = m0 + wm * 32 * MSUB + ms * 32;
#if defined(TILE_TRACE_BUILD) && defined(TILE_TRACE_NO_HP)
    // timing-only bound (tools/tile_trace.py, TRACE_FLAGS=-DTILE_TRACE_NO_HP; wrong results): the
    // epilogue WITHOUT its re-read of the previous states — what capturing them from the h
    // phase's LDS tiles could save at most
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hp_all[ms][r] = 0.f;
    } else
#endif
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + acc_row(r, lane);
        hp_all[ms][r] = hs[(off_prev + (m < S_t ? m : (S_t - 1))) * H + uc];
      }
    } else if (h0_rows != nullptr) {
      rowaddr_t hrow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + acc_row(r, lane);
        hrow[r] = h0_rows[m < S_t ? m : (S_t - 1)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) hp_all[ms][r] = reinterpret_cast<const float*>(hrow[r])[uc];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) hp_all[ms][r] = 0.f;
    }
  }
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) {
    const int mrow0 = m0 + wm * 32 * MSUB + ms * 32;
    float hn[16];
    const float (&hp)[16] = hp_all[ms];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mrow0 + acc_row(r, lane);
      const float rg = sigmoidf_(acc[ms][0][r] + b_r);
      const float zg = sigmoidf_(acc[ms][1][r] + b_z);
      const float ghn = acc[ms][3][r] + b_hn;
      const float ng = tanhf_(acc[ms][2][r] + b_in + rg * ghn);
      hn[r] = (1.0f - zg) * ng + zg * hp[r];
      if (BF3) {
        // the state once more in pre-split form for the next step's / the attention's A operand:
        // units u, u+1 sit in neighbouring lanes; even lanes store the (hi, lo) bf16 pairs
        const float other = __shfl_xor(hn[r], 1, 64);
        if (uv && m < S_t && (lane & 1) == 0) {
          const float o1 = (u + 1 < H) ? other : 0.f;
          const uint32_t hi = pack_bf16(hn[r], o1);
          const float f0 = __uint_as_float(hi << 16), f1 = __uint_as_float(hi & 0xffff0000u);
          const uint32_t lo = pack_bf16(hn[r] - f0, o1 - f1);
          uint32_t* dst = reinterpret_cast<uint32_t*>(p.hs_s) + (off_cur + m) * split_ld(H) +
                          (u >> 4) * 16 + ((u & 15) >> 1);
          dst[0] = hi;
          dst[8] = lo;
        }
      }
      if (uv && m < S_t) {
        if (CHAIN)   // read by the next step's tiles on other XCDs: past this XCD's L2 (sc1)
          __hip_atomic_store(&hs[(off_cur + m) * H + u], hn[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          hs[(off_cur + m) * H + u] = hn[r];
        if (gates != nullptr) {
          float* gp = gates + (off_cur + m) * 4 * H + u;
          gp[0] = rg;
          gp[H] = zg;
          gp[2 * H] = ng;
          gp[3 * H] = ghn;
        }
      }
    }
#ifdef TILE_TRACE_BUILD
    if (ms == MSUB - 1) {   // stores of the state drained: what the slot's successor waits for
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      TRACE_MARK(4);
    }
#endif
    if (pool_mode == CMHSE_POOL_ATTN) continue;   // pooled by attn_energy / attn_pool from hs

    // pooling fused into the step: index loads four rows at a time, then the dependent accesses
#pragma unroll
    for (int r4 = 0; r4 < 16; r4 += 4) {
      int orow[4], len[4];
      float cur[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mrow0 + acc_row(r4 + i, lane);
        const int mc = m < S_t ? m : (S_t - 1);
        orow[i] = out_row[mc];
        len[i] = (pool_mode == CMHSE_POOL_LAST) ? lens[mc] : 0;
      }
      if (pool_mode == CMHSE_POOL_MAX && t > 0) {
        // (CHAIN: the running maximum was written by the previous step's tile, on another CU — an
        // agent-scope load, which neither this CU's L1 nor a non-coherent L2 serves)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          cur[i] = CHAIN ? __hip_atomic_load(&out[static_cast<int64_t>(orow[i]) * H + uc], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)
                         : out[static_cast<int64_t>(orow[i]) * H + uc];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mrow0 + acc_row(r4 + i, lane);
        if (!(uv && m < S_t)) continue;
        const float v = hn[r4 + i];
        if (pool_mode == CMHSE_POOL_MAX) {
          if (t == 0 || v > cur[i]) {  // strict '>': the first maximum wins, like max_pool1d
            if (CHAIN)
              __hip_atomic_store(&out[static_cast<int64_t>(orow[i]) * H + u], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
              out[static_cast<int64_t>(orow[i]) * H + u] = v;
            if (argmax != nullptr) argmax[static_cast<int64_t>(m) * H + u] = t;
          }
        } else if (pool_mode == CMHSE_POOL_LAST) {
          if (t == len[i] - 1) out[static_cast<int64_t>(orow[i]) * H + u] = v;
        } else {  // CMHSE_POOL_ALL
          out[(static_cast<int64_t>(orow[i]) + t) * H + u] = v;
        }
      }
    }
  }
  if (CHAIN) {
    __builtin_amdgcn_s_waitcnt(0);   // this wave's write-through state stores have been performed
    flag_signal(dep.done);
  }
}

template <bool VEC, int MSUB, bool BF3>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSUB == 1 ? 3 : 2)))
void gru_step_kernel(const GruStepGroup grp) {
  unsigned wg;
  const GruStepParams& p = grp.j[group_job(grp, &wg)];
  ChainDep none;
  none.wait = nullptr;
  none.need = 0;
  none.done = nullptr;
  gru_step_tile<VEC, MSUB, BF3, false>(p, wg, p.t, p.S_t, p.off_prev, p.off_cur, none);
}

// ---------------------------------------------------------------------------------------------
// attention energies: e_part[nt][row] = sum_{n in N tile nt} w_att[n] * tanh(W_lin[n,:] . h_row + b)
// ---------------------------------------------------------------------------------------------
struct AttnEnergyParams {
  const float* hs_s;   // bf16x3: pre-split hidden states (rows of split_ld(H) units) or NULL
  const float* hs;     // [rows, H]
  const float* w_lin;  // [H, H]
  const float* w_lin_s;  // bf16x3 pre-split copy or NULL
  const float* b_lin;
  const float* w_att;
  float* e_part;  // [n_tiles, rows]
  float* v;       // [rows, H] tanh(W_lin h + b) kept for the backward pass, or NULL
  int64_t rows;   // all packed rows (stride of e_part)
  int64_t row_begin, row_end;   // the rows this launch computes
  int32_t H, n_tiles;
};


// One tile: packed rows [m0, m0 + 64 MSUB) (those below row_end) x columns [256 nt, 256 nt + 256).
// The body of attn_energy_kernel, and a task of the step chain (gru_step_chain_kernel).  A row's
// result does not depend on the tile height or on which rows share its tile.
template <bool VEC, int MSUB, bool BF3, bool ASPLIT>
__device__ __forceinline__ void attn_energy_tile(const AttnEnergyParams& p, const int nt, const int64_t m0,
                                                 const int64_t row_end) {
  constexpr int BM = 64 * MSUB, BN = kAttBN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = nt * BN;
  const int srow = tid >> 2;
  const int H = p.H;

  rowaddr_t ar[BM / 64];
  rowaddr_t br[BN / 64];
  bool av[BM / 64], bv[BN / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i) {
    const int64_t m = m0 + srow + 64 * i;
    av[i] = m < row_end;
    ar[i] = ASPLIT ? row_addr(p.hs_s + (av[i] ? m : (row_end - 1)) * split_ld(H))
                   : row_addr(p.hs + (av[i] ? m : (row_end - 1)) * H);
  }
#pragma unroll
  for (int i = 0; i < BN / 64; ++i) {
    const int n = n0 + srow + 64 * i;
    bv[i] = n < H;
    br[i] = BF3 ? row_addr(p.w_lin_s + static_cast<int64_t>(bv[i] ? n : (H - 1)) * split_ld(H))
                : row_addr(p.w_lin + static_cast<int64_t>(bv[i] ? n : (H - 1)) * H);
  }
  constexpr int NS = BN / 64;   // 32-column sub-tiles per wave
  f32x16 acc[MSUB][NS];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
    for (int a = 0; a < NS; ++a) acc[ms][a] = zero16();
  int b_row0[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) b_row0[ns] = wn * (BN / 2) + 32 * ns;
  if (BF3)
    nt_phase_bf3<BM, BN, MSUB, NS, NS, NS - 1, ASPLIT>(smem, ar, av, br, bv, H, wm * 32 * MSUB, b_row0, acc);
  else
    nt_phase<BM, BN, MSUB, NS, NS, NS - 1, VEC>(smem, ar, av, br, bv, H, wm * 32 * MSUB, b_row0, acc);

  // epilogue: per-row partial dot over this wave's BN/2 columns, then the two N-waves via LDS
  float wa[NS], bl[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int n = n0 + b_row0[ns] + acc_col(lane);
    wa[ns] = (n < H) ? p.w_att[n] : 0.f;
    bl[ns] = (n < H) ? p.b_lin[n] : 0.f;
  }
  float* red = smem;  // [2 (wn)][BM]; main loop ended with a barrier
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = 0.f;
      const int64_t vm = m0 + wm * 32 * MSUB + ms * 32 + acc_row(r, lane);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const float tv = tanhf_(acc[ms][ns][r] + bl[ns]);
        s += wa[ns] * tv;
        const int n = n0 + b_row0[ns] + acc_col(lane);
        if (p.v != nullptr && vm < row_end && n < H) p.v[vm * H + n] = tv;
      }
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
      if ((lane & 31) == 0) red[wn * BM + wm * 32 * MSUB + ms * 32 + acc_row(r, lane)] = s;
    }
  }
  __syncthreads();
  if (tid < BM) {
    const int64_t m = m0 + tid;
    if (m < row_end) p.e_part[static_cast<int64_t>(nt) * p.rows + m] = red[tid] + red[BM + tid];
  }
}

template <bool VEC, int MSUB, bool BF3, bool ASPLIT = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSUB == 1 ? 3 : 2)))
void attn_energy_kernel(const AttnEnergyParams p) {
  attn_energy_tile<VEC, MSUB, BF3, ASPLIT>(p, static_cast<int>(blockIdx.x % p.n_tiles),
                                           p.row_begin + static_cast<int64_t>(blockIdx.x / p.n_tiles) * (64 * MSUB),
                                           p.row_end);
}

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

// ---------------------------------------------------------------------------------------------
// Latency-shaped GRU step for small active sets (training batches, the long ragged tails of
// paragraphs): with S_t <= ~1k sequences the 64 x 64 tile above fills only part of the chip and every
// launch costs one full K loop (~100 us).  Here a workgroup owns 32 sequences x 8 hidden units:
//   * ONE MFMA per k-step computes all three gates of those 8 units: the 32 B columns of
//     v_mfma_f32_32x32x2_f32 are [r x8 | z x8 | n x8 | 8 unused];
//   * the x phase and the h phase accumulate into two separate 32x32 accumulators (the n gate
//     needs W_in x and W_hn h apart), so there are 2 x 16 accumulator registers per lane;
//   * the 4 waves split K four ways (wave w takes k-blocks w, w+4, ...), operand fragments go
//     global -> registers directly in MFMA layout through a 4-deep register ring (no LDS, no
//     barrier in the loop), and the partial tiles meet in LDS in a fixed order (deterministic);
//   * H/8 x ceil(S_t/32) workgroups: 128 even for a single active sequence at H = 1024.
// ---------------------------------------------------------------------------------------------
constexpr int kTinyBM = 32;
constexpr int kTinyBU = 8;

// NW = waves per workgroup splitting K: 4, or 8 when so few sequences are active that the launch
// is a pure latency chain (half the MFMA chain per wave, twice the waves on an under-filled chip).
template <bool VEC, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gru_step_tiny_kernel(const GruStepGroup grp) {
  CHAIN_WAVE_PRIORITY();
  constexpr int BM = kTinyBM, BU = kTinyBU;
  unsigned wg;
  const GruStepParams& p = grp.j[group_job(grp, &wg)];
  __shared__ float red[NW][2][16][64];  // [wave][x|h accumulator][register][lane], 8 KB per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u_tiles = (p.H + BU - 1) / BU;
  const int u0 = (wg % u_tiles) * BU;  // unit tile fastest: b, b+8 share an XCD's L2
  const int m0 = (wg / u_tiles) * BM;
  const int I = p.I, H = p.H;
  const int row = lane & 31, hi = lane >> 5;

  // A fragment row of this lane: sequence m0 + row (clamped; rows are independent, and rows past
  // S_t are never stored)
  const int m = m0 + row;
  const int mc = (m < p.S_t) ? m : (p.S_t - 1);
  rowaddr_t ax, ah;
  if (p.tok_rows != nullptr) {
    long long tok = reinterpret_cast<const long long*>(p.tok_rows[mc])[p.t];
    tok = tok < 0 ? 0 : (tok >= p.vocab ? p.vocab - 1 : tok);
    ax = row_addr(p.emb + tok * I);
  } else {
    ax = p.x_rows[mc] + static_cast<rowaddr_t>(p.t) * p.x_step * 4u;
  }
  const bool have_h = (p.t > 0) || (p.h0_rows != nullptr);
  if (p.t > 0)
    ah = row_addr(p.hs + (p.off_prev + mc) * H);
  else if (p.h0_rows != nullptr)
    ah = p.h0_rows[mc];
  else
    ah = row_addr(p.w_hh);
  // B fragment row of this lane: column `row` of the MFMA = gate row>>3 of unit u0 + (row&7)
  const int g = row >> 3, uu = u0 + (row & 7);
  const bool bvalid = (g < 3) && (uu < H);
  const int gc = (g < 3) ? g : 2, uc = (uu < H) ? uu : (H - 1);
  const rowaddr_t bx = row_addr(p.w_ih + (static_cast<int64_t>(gc) * H + uc) * I);
  const rowaddr_t bh = row_addr(p.w_hh + (static_cast<int64_t>(gc) * H + uc) * H);

  f32x16 acc_x = zero16(), acc_h = zero16();
  tiny_phase<VEC, NW>(ax, bx, bvalid, I, wave, hi, acc_x);
  if (have_h) tiny_phase<VEC, NW>(ah, bh, bvalid, H, wave, hi, acc_h);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[wave][0][r][lane] = acc_x[r];
    red[wave][1][r][lane] = acc_h[r];
  }
  __syncthreads();

  // epilogue: one (sequence, unit) per thread; its three gate columns sit in lanes col, col+8,
  // col+16 of the half-wave that owns the row
  const int er = tid >> 3, eu = tid & 7;         // tile row 0..31, unit 0..7
  const int em = m0 + er, u = u0 + eu;
  if (tid >= 256 || em >= p.S_t || u >= H) return;   // (with NW = 8 the upper four waves only split K)
  const int reg = (er & 3) | ((er >> 3) << 2);
  const int lbase = 32 * ((er >> 2) & 1) + eu;
  float xr = 0.f, xz = 0.f, xn = 0.f, hr = 0.f, hz = 0.f, hn_ = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    xr += red[w][0][reg][lbase];
    xz += red[w][0][reg][lbase + 8];
    xn += red[w][0][reg][lbase + 16];
    hr += red[w][1][reg][lbase];
    hz += red[w][1][reg][lbase + 8];
    hn_ += red[w][1][reg][lbase + 16];
  }
  float hp = 0.f;
  if (p.t > 0)
    hp = p.hs[(p.off_prev + em) * H + u];
  else if (p.h0_rows != nullptr)
    hp = reinterpret_cast<const float*>(p.h0_rows[em])[u];
  const float rg = sigmoidf_(xr + hr + p.b_ih[u] + p.b_hh[u]);
  const float zg = sigmoidf_(xz + hz + p.b_ih[H + u] + p.b_hh[H + u]);
  const float ghn = hn_ + p.b_hh[2 * H + u];
  const float ng = tanhf_(xn + p.b_ih[2 * H + u] + rg * ghn);
  const float hn = (1.0f - zg) * ng + zg * hp;
  p.hs[(p.off_cur + em) * H + u] = hn;
  if (p.gates != nullptr) {
    float* gp = p.gates + (p.off_cur + em) * 4 * H + u;
    gp[0] = rg;
    gp[H] = zg;
    gp[2 * H] = ng;
    gp[3 * H] = ghn;
  }
  if (p.pool_mode == CMHSE_POOL_MAX) {
    float* o = p.out + static_cast<int64_t>(p.out_row[em]) * H + u;
    if (p.t == 0 || hn > *o) {
      *o = hn;
      if (p.argmax != nullptr) p.argmax[static_cast<int64_t>(em) * H + u] = p.t;
    }
  } else if (p.pool_mode == CMHSE_POOL_LAST) {
    if (p.t == p.lens[em] - 1) p.out[static_cast<int64_t>(p.out_row[em]) * H + u] = hn;
  } else if (p.pool_mode == CMHSE_POOL_ALL) {
    p.out[(static_cast<int64_t>(p.out_row[em]) + p.t) * H + u] = hn;
  }
}

// ---------------------------------------------------------------------------------------------
// Mid-size GRU step: 1 <= S_t <= ~1k active sequences (every step of a training batch, the level-2
// encoders, the long few-sequence tails of paragraphs).  Such a step is one [S_t, K] x [K, 3H]
// product with S_t far too small to fill 256 CUs from LDS-tiled 64-row tiles, and the 32 x 8-unit
// tiles of gru_step_tiny_kernel pull every operand row through L2 once per tile (~220 MB per step
// at S_t = 152: that kernel is L2-bandwidth-bound, not latency-bound).  Two changes:
//   * the input projection x_t W_ih^T has no time dependence: for all these steps together it is
//     ONE well-shaped GEMM (xproj_kernel, tiled like the attention projection) into gx[rows, 3H];
//     the sequential part keeps only K = H;
//   * tile = 32 (or 16) sequences x 16, 8 or 4 hidden units x {r, z, n}: blocks of
//     v_mfma_f32_16x16x4_f32, 8 waves split K, operands global -> registers in MFMA layout through
//     a ring of ONE 128-byte line pair per wave, fixed-order LDS combine, the epilogue's operands
//     requested before the K loop; H/BU x ceil(S_t/32) workgroups of 512 threads.
// In-kernel stamps (tools/mid_trace.py): the loop is bound by how fast ONE CU pulls its operands
// through L1 (a 32 x 16 tile needs 320 KB; 40-50 GB/s per CU for a plain stream of an L2-resident
// slice, tools/microbench/weights_reread.hip).  Measured (tools/step_sweep.py, us per step at
// H = 1024): deeper rings are SLOWER (4 waves x 4 blocks in flight: 22.0 at S_t = 96; 8 x 2: 17.2;
// 8 x 4: 19.9; 8 x 8 on the 4-unit tile: 21.5 against 9.5 at S_t = 16), with non-temporal loads
// too (slower still at every depth: the second half of a line does hit L1): the loop wants many
// waves with little in flight each.  Also measured (late round 3): the BPTT step's form — the
// product as 32 x 128 LDS-staged tiles with K split over the grid (bwd_rec_part_kernel on
// h_{t-1} . W_hh^T) plus a gates kernel, two launches — for the steps of a training chain with
// more than 32 sequences: correct, and 0.4 ms per training step SLOWER (ICEP 9.19 -> 9.60 ms, C3D
// 7.88 -> 8.33): with K = H instead of 3H there are 96 tiles of two short slices, and the second
// launch costs more than the better-coalesced loads save (DiDeMo, ~210 sequences at every step:
// 10.82 -> 11.19 ms).  And 16 waves on 16 K slices (1024
// threads) instead of 8 on 8: 21.0 -> 21.6 us at S_t = 117, 30.9 -> 34.1 at 152.
// ---------------------------------------------------------------------------------------------
// MB = 16-row blocks of sequences per workgroup: 2 (32 sequences), or 1 when at most 16 are active.
// BU = hidden units per workgroup (16, 8 or 4).  The 3 BU gate columns (gate-major: column
// f = gate * BU + unit) fill NB = ceil(3 BU / 16) MFMA column blocks.  A step with few sequences
// has only H / 16 x ceil(S_t / 32) tiles of 16 units — 64 workgroups at S_t <= 32, H = 1024, each
// pulling 320 KB through ONE CU's L2 port (~40-50 GB/s, tools/microbench/weights_reread.hip) while
// three quarters of the chip idle; narrower unit tiles spread the same W_hh over up to 256 CUs
// (176 KB per workgroup at BU = 4: the 32 h rows are then the larger part).  mid_units() picks BU.
// Waves per workgroup (NW, splitting K) and 16-k blocks in flight per wave (D).  8 x 2 is the
// fastest shape for a chain that has the chip to itself (a training step's towers, the level-2
// encoders).  A chain that runs BESIDE other kernels — the few-sequence tail of the text encoder
// on the side stream while the visual encoder still launches LDS-tiled steps or runs its attention
// pass — uses 4 waves: a 512-thread workgroup needs two free wave slots on every SIMD of one CU at
// once and starves among 256-thread workgroups that refill slots one by one (615-video share of
// the split: 50.2 ms per pass with 8 waves, 42.5 with 4).
constexpr int kMidRing = 2;   // 16-k blocks in flight per wave (4 and 8 measured slower, see above)

// K is always cut into kMidSlices = 8 slices with one accumulator each, combined in slice order:
// with 8 waves every wave owns one slice, with 4 waves wave w runs slices w and w + 4 one after
// the other — the same arithmetic, so both shapes give bit-identical results and the choice
// between them is free to follow the schedule.
constexpr int kMidSlices = 8;

template <int MB, int BU, int NW>
__global__ __launch_bounds__(64 * NW) void gru_step_mid_kernel(const GruStepGroup grp) {
  CHAIN_WAVE_PRIORITY();
  constexpr int BM = 16 * MB, NB = (3 * BU + 15) / 16, VS = kMidSlices / NW;
  static_assert(NW * VS == kMidSlices, "4 or 8 waves");
  constexpr int OUTS = BM * BU, NOUT = (OUTS + 64 * NW - 1) / (64 * NW);   // outputs (per thread)
  unsigned wg;
  const GruStepParams& p = grp.j[group_job(grp, &wg)];
  __shared__ f32x4v red[kMidSlices][MB * NB][64];   // [K slice][M block x column block][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H;
  const int u_tiles = (H + BU - 1) / BU;
  const int u0 = (wg % u_tiles) * BU;    // unit tile fastest: b, b+8 share an XCD's L2
  const int m0 = (wg / u_tiles) * BM;
  const int r16 = lane & 15, kq = lane >> 4;
  const bool have_h = (p.t > 0) || (p.h0_rows != nullptr);
#ifdef TILE_TRACE_BUILD
  // tools/mid_trace.py: stamps of step t, workgroup wg at g_trace[(t * gridDim.x + blockIdx.x) * 8 + i]
#define MID_MARK(i)                                                                       \
  do {                                                                                    \
    if (threadIdx.x == 0 && g_trace)                                                      \
      g_trace[(static_cast<size_t>(p.t) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define MID_MARK(i) do {} while (0)
#endif
  MID_MARK(0);
  // The epilogue's own operands do not depend on the K loop: request them first (branch-free,
  // clamped), so their memory round trip hides under it.  Output o of this thread: tile row
  // er = o / BU, unit eu = o % BU, o = tid + 256 q.
  float e_gx[NOUT][3], e_hp[NOUT], e_b[NOUT][4];
#pragma unroll
  for (int q = 0; q < NOUT; ++q) {
    const int o = tid + 64 * NW * q;
    const int em = m0 + (o / BU) % BM, u = u0 + (o % BU);
    const int emc = (em < p.S_t) ? em : (p.S_t - 1), uc = (u < H) ? u : (H - 1);
    const int64_t gxrow = p.gx_per_seq ? static_cast<int64_t>(emc) : (p.off_cur + emc - p.gx_p0);
    const float* gxr = p.gx + gxrow * 3 * H;
    e_gx[q][0] = gxr[uc];
    e_gx[q][1] = gxr[H + uc];
    e_gx[q][2] = gxr[2 * H + uc];
    if (p.t > 0)
      e_hp[q] = p.hs[(p.off_prev + emc) * H + uc];
    else if (p.h0_rows != nullptr)
      e_hp[q] = reinterpret_cast<const float*>(p.h0_rows[emc])[uc];
    else
      e_hp[q] = 0.f;
    e_b[q][0] = p.b_ih[uc] + p.b_hh[uc];
    e_b[q][1] = p.b_ih[H + uc] + p.b_hh[H + uc];
    e_b[q][2] = p.b_ih[2 * H + uc];
    e_b[q][3] = p.b_hh[2 * H + uc];
  }
  if (have_h) {
    rowaddr_t arow[MB], brow[NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = m0 + 16 * mb + r16;
      const int mc = (m < p.S_t) ? m : (p.S_t - 1);   // rows past S_t are never stored
      arow[mb] = (p.t > 0) ? row_addr(p.hs + (p.off_prev + mc) * H) : p.h0_rows[mc];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      // column 16 j + r16 of the gate-major tile; columns past 3 BU (and units past H) compute on
      // a clamped row and are never read back
      const int fc = (16 * j + r16 < 3 * BU) ? (16 * j + r16) : (3 * BU - 1);
      const int uu = u0 + fc % BU, uc = (uu < H) ? uu : (H - 1);
      brow[j] = row_addr(p.w_hh + (static_cast<int64_t>(fc / BU) * H + uc) * H);
    }
    MID_MARK(1);
#pragma unroll
    for (int v = 0; v < VS; ++v) {
      const int slice = wave + NW * v;
      f32x4v acc[MB][NB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[mb][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      mid_phase<MB, NB, kMidSlices, kMidRing>(arow, brow, H, slice, kq, acc);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NB; ++j) red[slice][mb * NB + j][lane] = acc[mb][j];
    }
    MID_MARK(2);
    __syncthreads();
    MID_MARK(3);
  }

  // epilogue.  Element (row r, col c) of a 16x16 block sits in lane (r >> 2) * 16 + c, register r & 3.
#pragma unroll
  for (int q = 0; q < NOUT; ++q) {
    const int o = tid + 64 * NW * q;
    if (o >= OUTS) continue;
    const int er = o / BU, eu = o % BU;
    const int em = m0 + er, u = u0 + eu;
    if (em >= p.S_t || u >= H) continue;
    float hg[3] = {0.f, 0.f, 0.f};
    if (have_h) {
      const int mb = er >> 4, rr = er & 15, reg = rr & 3;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int fc = g * BU + eu;
        const int sl = (rr >> 2) * 16 + (fc & 15);
#pragma unroll
        for (int w = 0; w < kMidSlices; ++w)
          hg[g] += reinterpret_cast<const float*>(&red[w][mb * NB + (fc >> 4)][sl])[reg];
      }
    }
    const float hr = hg[0], hz = hg[1], hn_ = hg[2];
    const float rg = sigmoidf_(e_gx[q][0] + hr + e_b[q][0]);
    const float zg = sigmoidf_(e_gx[q][1] + hz + e_b[q][1]);
    const float ghn = hn_ + e_b[q][3];
    const float ng = tanhf_(e_gx[q][2] + e_b[q][2] + rg * ghn);
    const float hn = (1.0f - zg) * ng + zg * e_hp[q];
    p.hs[(p.off_cur + em) * H + u] = hn;
    if (p.gates != nullptr) {
      float* gp = p.gates + (p.off_cur + em) * 4 * H + u;
      gp[0] = rg;
      gp[H] = zg;
      gp[2 * H] = ng;
      gp[3 * H] = ghn;
    }
    if (p.pool_mode == CMHSE_POOL_MAX) {
      float* op = p.out + static_cast<int64_t>(p.out_row[em]) * H + u;
      if (p.t == 0 || hn > *op) {
        *op = hn;
        if (p.argmax != nullptr) p.argmax[static_cast<int64_t>(em) * H + u] = p.t;
      }
    } else if (p.pool_mode == CMHSE_POOL_LAST) {
      if (p.t == p.lens[em] - 1) p.out[static_cast<int64_t>(p.out_row[em]) * H + u] = hn;
    } else if (p.pool_mode == CMHSE_POOL_ALL) {
      p.out[(static_cast<int64_t>(p.out_row[em]) + p.t) * H + u] = hn;
    }
  }
#ifdef TILE_TRACE_BUILD
  __builtin_amdgcn_s_waitcnt(0);
  MID_MARK(5);
#endif
}

// ---------------------------------------------------------------------------------------------
// The few-sequence TAIL of a training chain as ONE resident kernel (the forward twin of
// gru_bwd_tail_kernel, bwd.hip — read its header for the why and for the coherence argument): the
// steps t >= t_lo with at most 32 active sequences, each a 16- or 32-sequence x 16-unit tile per
// workgroup.  The
// workgroup's 48 rows of W_hh (3 gates x 16 units) sit in registers in mid_phase's operand layout;
// per step only h_{t-1} crosses workgroups: written through (agent-scope stores — every hs row is
// written once, to an address nobody read in this kernel), read past the non-coherent L2s
// (sc1 buffer loads) behind the step's grid barrier.  Block ownership, accumulation and combine
// order are gru_step_mid_kernel<1, 16, 8>'s.
// ---------------------------------------------------------------------------------------------
struct FwdTailParams {
  GruStepParams p;           // as for a step of the chain; t / S_t / off_* are derived per step
  const int32_t* step_off;   // device [Tmax + 1]
  GridSync sync;             // grid barrier words (zeroed by the caller)
  int32_t t_lo, t_hi;        // steps t_lo >= 1 ... t_hi = Tmax - 1
};

constexpr int kFwdTailMaxSeqs = 32;             // two 16-row blocks per workgroup

template <int KBMAX, int MB>
__global__ __launch_bounds__(512) void gru_fwd_tail_kernel(const FwdTailParams q) {
  CHAIN_WAVE_PRIORITY();
  constexpr int NW = 8, NB = 3, BU = 16;
  const GruStepParams& p = q.p;
  __shared__ f32x4v red[NW][MB * NB][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H;
  const int u0 = blockIdx.x * BU;
  const int r16 = lane & 15, kq = lane >> 4;
  const int nkb = H / 16;
  auto block_of = [&](int i) { return (i >> 1) * 2 * NW + 2 * wave + (i & 1); };   // mid_phase's ownership
  int nmine = 0;
  while (nmine < KBMAX && block_of(nmine) < nkb) ++nmine;
  // B operand: column 16 j + r16 of the gate-major tile = row (gate j, unit u0 + r16) of W_hh
  float4 wreg[NB][KBMAX];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int uu = u0 + r16, uc = (uu < H) ? uu : (H - 1);
    const float* brow = p.w_hh + (static_cast<int64_t>(j) * H + uc) * H;
#pragma unroll
    for (int i = 0; i < KBMAX; ++i)
      wreg[j][i] = (i < nmine) ? *reinterpret_cast<const float4*>(brow + block_of(i) * 16 + 4 * kq)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the output this thread owns (16 MB x 16 of them: threads 0..255 at MB = 1, all 512 at MB = 2):
  // tile row er = sorted sequence, unit u
  const int er = tid >> 4, eu = tid & 15;
  const int u = u0 + eu;
  const bool owner = er < 16 * MB && u < H;
  float e_b[4] = {0.f, 0.f, 0.f, 0.f};
  if (owner) {
    e_b[0] = p.b_ih[u] + p.b_hh[u];
    e_b[1] = p.b_ih[H + u] + p.b_hh[H + u];
    e_b[2] = p.b_ih[2 * H + u];
    e_b[3] = p.b_hh[2 * H + u];
  }
  float hprev = 0.f;
  {
    const int off_prev = q.step_off[q.t_lo - 1];
    const int S_lo = q.step_off[q.t_lo + 1] - q.step_off[q.t_lo];
    if (owner && er < S_lo) hprev = p.hs[(static_cast<int64_t>(off_prev) + er) * H + u];
  }
  unsigned arrivals = 0;
  for (int t = q.t_lo; t <= q.t_hi; ++t) {
    const int off_cur = q.step_off[t], off_prev = q.step_off[t - 1];
    const int S_t = q.step_off[t + 1] - off_cur;
    // the epilogue's own operands do not depend on the chain: request them first
    float e_gx[3] = {0.f, 0.f, 0.f};
    if (owner && er < S_t) {
      const int64_t gxrow = p.gx_per_seq ? static_cast<int64_t>(er)
                                         : (static_cast<int64_t>(off_cur) + er - p.gx_p0);
      const float* gxr = p.gx + gxrow * 3 * H;
      e_gx[0] = gxr[u];
      e_gx[1] = gxr[H + u];
      e_gx[2] = gxr[2 * H + u];
    }
    {
      // A operand: the rows of step t - 1 (the previous kernel's for t = t_lo, else published by
      // every workgroup before the barrier at the end of the previous trip)
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          p.hs + static_cast<int64_t>(off_prev) * H, 0, 0x7fffffff, 0x00020000);
      typedef int i32x4v __attribute__((ext_vector_type(4)));
      i32x4v areg[MB][KBMAX];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = 16 * mb + r16;
        const int row_b = ((m < S_t) ? m : (S_t - 1)) * H * 4;
#pragma unroll
        for (int i = 0; i < KBMAX; ++i)
          if (i < nmine && 16 * mb < S_t)
            areg[mb][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, row_b + (block_of(i) * 16 + 4 * kq) * 4, 0, 16);
      }
      f32x4v acc[MB][NB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[mb][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KBMAX; ++i) {
        if (i >= nmine) continue;   // wave-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            if (16 * mb >= S_t) continue;   // (workgroup-uniform) an empty row block
            const int ai = (c == 0) ? areg[mb][i].x : (c == 1) ? areg[mb][i].y : (c == 2) ? areg[mb][i].z : areg[mb][i].w;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const float bv = (c == 0) ? wreg[j][i].x : (c == 1) ? wreg[j][i].y : (c == 2) ? wreg[j][i].z : wreg[j][i].w;
              acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(ai), bv, acc[mb][j], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NB; ++j) red[wave][mb * NB + j][lane] = acc[mb][j];
      __syncthreads();
    }
    if (owner && er < S_t) {
      const int mb = er >> 4, rr = er & 15;
      const int sl = (rr >> 2) * 16 + eu, reg = rr & 3;
      float hg[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int w = 0; w < NW; ++w) hg[g] += reinterpret_cast<const float*>(&red[w][mb * NB + g][sl])[reg];
      const float rg = sigmoidf_(e_gx[0] + hg[0] + e_b[0]);
      const float zg = sigmoidf_(e_gx[1] + hg[1] + e_b[1]);
      const float ghn = hg[2] + e_b[3];
      const float ng = tanhf_(e_gx[2] + e_b[2] + rg * ghn);
      const float hn = (1.0f - zg) * ng + zg * hprev;
      hprev = hn;
      const int64_t row = static_cast<int64_t>(off_cur) + er;
      // next step's A operand, in every workgroup: write through to where all XCDs see it
      __hip_atomic_store(&p.hs[row * H + u], hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.gates != nullptr) {
        float* gp = p.gates + row * 4 * H + u;
        gp[0] = rg;
        gp[H] = zg;
        gp[2 * H] = ng;
        gp[3 * H] = ghn;
      }
      if (p.pool_mode == CMHSE_POOL_MAX) {
        float* op = p.out + static_cast<int64_t>(p.out_row[er]) * H + u;
        if (hn > *op) {      // (t >= 1 here: the running maximum exists)
          *op = hn;
          if (p.argmax != nullptr) p.argmax[static_cast<int64_t>(er) * H + u] = t;
        }
      } else if (p.pool_mode == CMHSE_POOL_LAST) {
        if (t == p.lens[er] - 1) p.out[static_cast<int64_t>(p.out_row[er]) * H + u] = hn;
      } else if (p.pool_mode == CMHSE_POOL_ALL) {
        p.out[(static_cast<int64_t>(p.out_row[er]) + t) * H + u] = hn;
      }
    }
    if (t == q.t_hi) break;
    __builtin_amdgcn_s_waitcnt(0);
    arrivals += gridDim.x;
    if (!grid_sync_wait(q.sync, arrivals)) return;
  }
}

// Hoisted input projection of the mid-size steps: gx[m][n] = sum_k x_row(m)[k] W_ih[n][k] for the
// packed rows p0 + m of steps >= t_first (or, for a time-constant input, for the sequences
// themselves), 64 x 192 tiles on the shared exact-fp32 NT tile loop.
struct XprojParams {
  const uint64_t* x_rows;
  const uint64_t* tok_rows;
  const float* emb;
  const int32_t* step_off;
  const float* w_ih;
  float* gx;
  int64_t p0, rows;      // gx row m is packed row p0 + m; this launch computes gx rows [m_begin, rows)
  int64_t m_begin;
  int32_t I, N, vocab, x_step, Tmax, t_first, n_tiles, per_seq;
};

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3)))
void xproj_kernel(const XprojParams p) {
  constexpr int BM = 64, BN = 192, NS = 3;   // the step kernel's tile shape (3 x 32 columns per wave)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = (blockIdx.x % p.n_tiles) * BN;
  const int64_t m0 = p.m_begin + static_cast<int64_t>(blockIdx.x / p.n_tiles) * BM;
  const int srow = tid >> 2;
  rowaddr_t ar[1], br[BN / 64];
  bool av[1], bv[BN / 64];
  {
    int64_t m = m0 + srow;
    av[0] = m < p.rows;
    if (!av[0]) m = p.rows - 1;
    int t = 0;
    int64_t sidx = m;
    if (!p.per_seq) {
      // packed row -> (step, sorted sequence): the last step whose first row is <= the row
      const int64_t pr = p.p0 + m;
      int lo = p.t_first, hi = p.Tmax - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (static_cast<int64_t>(p.step_off[mid]) <= pr) lo = mid; else hi = mid - 1;
      }
      t = lo;
      sidx = pr - p.step_off[t];
    }
    if (p.tok_rows != nullptr) {
      long long tok = reinterpret_cast<const long long*>(p.tok_rows[sidx])[t];
      tok = tok < 0 ? 0 : (tok >= p.vocab ? p.vocab - 1 : tok);
      ar[0] = row_addr(p.emb + tok * p.I);
    } else {
      ar[0] = p.x_rows[sidx] + static_cast<rowaddr_t>(t) * p.x_step * 4u;
    }
  }
#pragma unroll
  for (int i = 0; i < BN / 64; ++i) {
    const int n = n0 + srow + 64 * i;
    bv[i] = n < p.N;
    br[i] = row_addr(p.w_ih + static_cast<int64_t>(bv[i] ? n : (p.N - 1)) * p.I);
  }
  f32x16 acc[1][NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) acc[0][a] = zero16();
  int b_row0[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) b_row0[ns] = wn * (BN / 2) + 32 * ns;
  nt_phase<BM, BN, 1, NS, NS, NS - 1, true>(smem, ar, av, br, bv, p.I, wm * 32, b_row0, acc);
  // Epilogue.  48 dword stores per lane (one per accumulator element) made this kernel
  // store-ISSUE-bound (65 % MFMA-busy against 86 % for the step kernel on the same tile loop): the
  // tile goes through the now idle LDS, 32 rows at a time, and leaves as 16-byte stores of whole
  // row segments (6 per thread and half).
  constexpr int kLd = BN + 4;                  // row stride of the staging image, floats
  float* stage = smem;                         // 32 x 196 x 4 B = 25 KB of the 40 KB tile buffers
  const bool vec_out = (p.N % 4 == 0) && (n0 + BN <= p.N);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          stage[acc_row(r, lane) * kLd + b_row0[ns] + acc_col(lane)] = acc[0][ns][r];
    }
    __syncthreads();
    const int64_t mh = m0 + 32 * half;
    if (vec_out) {
#pragma unroll
      for (int i = 0; i < (32 * BN / 4) / kThreads; ++i) {
        const int idx = tid + kThreads * i;
        const int row = idx / (BN / 4), c4 = idx % (BN / 4);
        if (mh + row < p.rows)
          *reinterpret_cast<float4*>(p.gx + (mh + row) * p.N + n0 + 4 * c4) =
              *reinterpret_cast<const float4*>(stage + row * kLd + 4 * c4);
      }
    } else {
      for (int idx = tid; idx < 32 * BN; idx += kThreads) {
        const int row = idx / BN, c = idx % BN;
        if (mh + row < p.rows && n0 + c < p.N) p.gx[(mh + row) * p.N + n0 + c] = stage[row * kLd + c];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// attention pooling: out[s] = sum_t a_t h_t,  a_t = exp(e_t) [t < len] / (sum_t exp(e_t) + 1e-4)
// one workgroup per sequence; reads each hidden row once (HBM-bound).
// ---------------------------------------------------------------------------------------------
struct AttnPoolParams {
  const float* hs;
  const float* e_part;
  const int32_t* lens;
  const int32_t* out_row;
  const int32_t* step_off;
  float* out;
  int64_t rows;
  int32_t H, n_tiles;
};

__global__ __launch_bounds__(kThreads) void attn_pool_kernel(const AttnPoolParams p) {
  const int s = blockIdx.x;
  const int len = p.lens[s];
  const int tid = threadIdx.x;
  __shared__ float s_w[kThreads];
  __shared__ int64_t s_row[kThreads];
  __shared__ float s_den;
  // pass 1: denominator sum_t exp(e_t) + 1e-4 (each exp evaluated by exactly one thread)
  float part = 0.f;
  for (int t = tid; t < len; t += kThreads) {
    const int64_t row = static_cast<int64_t>(p.step_off[t]) + s;
    float e = 0.f;
    for (int q = 0; q < p.n_tiles; ++q) e += p.e_part[q * p.rows + row];
    part += expf(e);
  }
  s_w[tid] = part;
  __syncthreads();
  if (tid == 0) {
    float d = 0.f;
    for (int i = 0; i < kThreads; ++i) d += s_w[i];
    s_den = d + 0.0001f;
  }
  __syncthreads();
  const float den = s_den;
  const int H = p.H;
  float* o = p.out + static_cast<int64_t>(p.out_row[s]) * H;
  const bool vec = (H % 4 == 0) && aligned16(p.hs) && aligned16(o);
  // pass 2: weighted sum over the sequence's rows; the weights AND the packed row numbers of 256
  // steps at a time are staged in LDS, so the row loads of consecutive steps are independent of any
  // other global load and pipeline freely (HBM-bound: each hidden row is read once, 16 B per lane)
  for (int ub = 0; ub < H; ub += 4 * kThreads) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int u = ub + 4 * tid;
    for (int t0 = 0; t0 < len; t0 += kThreads) {
      __syncthreads();
      if (t0 + tid < len) {
        const int64_t row = static_cast<int64_t>(p.step_off[t0 + tid]) + s;
        float e = 0.f;
        for (int q = 0; q < p.n_tiles; ++q) e += p.e_part[q * p.rows + row];
        s_w[tid] = expf(e) / den;
        s_row[tid] = row;
      }
      __syncthreads();
      const int cnt = (len - t0 < kThreads) ? (len - t0) : kThreads;
      if (vec && u + 3 < H) {
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {
          const float4 h = *reinterpret_cast<const float4*>(p.hs + s_row[j] * H + u);
          const float wgt = s_w[j];
          a0 += wgt * h.x;
          a1 += wgt * h.y;
          a2 += wgt * h.z;
          a3 += wgt * h.w;
        }
      } else {
        for (int j = 0; j < cnt; ++j) {
          const float* hrow = p.hs + s_row[j] * H;
          const float wgt = s_w[j];
          if (u < H) a0 += wgt * hrow[u];
          if (u + 1 < H) a1 += wgt * hrow[u + 1];
          if (u + 2 < H) a2 += wgt * hrow[u + 2];
          if (u + 3 < H) a3 += wgt * hrow[u + 3];
        }
      }
    }
    if (vec && u + 3 < H) {
      *reinterpret_cast<float4*>(o + u) = make_float4(a0, a1, a2, a3);
    } else {
      if (u < H) o[u] = a0;
      if (u + 1 < H) o[u + 1] = a1;
      if (u + 2 < H) o[u + 2] = a2;
      if (u + 3 < H) o[u + 3] = a3;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// F.normalize: y = x / max(||x||_2, 1e-12), one workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void l2norm_rows_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, int cols,
                                                               int64_t ld) {
  const int64_t row = blockIdx.x;
  const float* xr = x + row * ld;
  float* yr = y + row * ld;
  float ss = 0.f;
  for (int c = threadIdx.x; c < cols; c += kThreads) {
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
  for (int c = threadIdx.x; c < cols; c += kThreads) yr[c] = xr[c] * inv;
}

// nn.Embedding lookup as a plain row gather (only used when the caller asks for the word tensor,
// model.py:94,98; the encoders fuse the lookup into their operand loads instead).
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ table,
                                                               const long long* __restrict__ ids,
                                                               float* __restrict__ out, int cols,
                                                               int vocab) {
  const int64_t r = blockIdx.x;
  long long id = ids[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* src = table + id * cols;
  float* dst = out + r * cols;
  for (int c = threadIdx.x; c < cols; c += kThreads) dst[c] = src[c];
}

// ---------------------------------------------------------------------------------------------
// Host -> HBM upload of time steps [t0, t1) of every still-active sequence, straight out of the
// loader's pinned host tensors (device-readable, zero-copy over PCIe): the unit the step pipeline
// consumes.  Padding rows (t >= len) are never read on the host side nor written here.
// A few dozen waves saturate PCIe (tools/microbench/h2d_chunked.hip), so the grid is small: the
// kernel runs beside the MFMA-bound step kernels and must not crowd their CUs.
// ---------------------------------------------------------------------------------------------
struct PullParams {
  const uint64_t* src_rows;   // [S] host (pinned) address of step 0 of sorted sequence s
  const uint64_t* dst_rows;   // [S] device address of step 0 of sorted sequence s
  const int32_t* lens;        // [S] non-increasing
  int32_t n_active, row_floats, t0, t1;
};

__global__ __launch_bounds__(kThreads) void pull_steps_kernel(const PullParams p) {
  const unsigned nthr = blockDim.x;
  for (int s = blockIdx.x; s < p.n_active; s += gridDim.x) {
    const int len = p.lens[s];
    const int te = (p.t1 < len) ? p.t1 : len;
    if (te <= p.t0) break;     // sorted longest first: every later sequence is shorter still
    const size_t off = static_cast<size_t>(p.t0) * p.row_floats * 4u;
    const size_t n = static_cast<size_t>(te - p.t0) * p.row_floats;
    const rowaddr_t src = p.src_rows[s] + off, dst = p.dst_rows[s] + off;
    if (((src | dst) & 15u) == 0 && (n & 3u) == 0) {
      const float4* sp = reinterpret_cast<const float4*>(src);
      float4* dp = reinterpret_cast<float4*>(dst);
      const size_t n4 = n >> 2;
      size_t i = threadIdx.x;
      // eight 16-byte PCIe reads in flight per lane: few waves must keep the link busy, because
      // every resident pull wave costs the MFMA-bound step kernel beside it a workgroup slot
      for (; i + 7 * nthr < n4; i += 8 * nthr) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = sp[i + q * nthr];
#pragma unroll
        for (int q = 0; q < 8; ++q) dp[i + q * nthr] = v[q];
      }
      for (; i < n4; i += nthr) dp[i] = sp[i];
    } else {
      const float* sp = reinterpret_cast<const float*>(src);
      float* dp = reinterpret_cast<float*>(dst);
      for (size_t i = threadIdx.x; i < n; i += nthr) dp[i] = sp[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// HBM -> host hand-over of finished embedding rows (the `.data.cpu()` of evaluation.py:120-125):
// a plain byte copy into page-locked, device-writable host memory, done by a FEW single-wave
// workgroups.  The runtime's own device-to-host copy of this size is a chip-wide blit kernel: beside
// the level-2 step chain (which needs its workgroups resident together) it held that chain back by
// 2 ms and the launches queued behind it by another (profiles/r06_api_path.txt).  PCIe writes are
// posted: a few dozen waves keep the link full.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void push_bytes_kernel(const float4* __restrict__ src,
                                                        float4* __restrict__ dst, size_t n16,
                                                        const unsigned char* __restrict__ src_tail,
                                                        unsigned char* __restrict__ dst_tail, int tail) {
  const size_t nthr = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 7 * nthr < n16; i += 8 * nthr) {
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[i + q * nthr];
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[i + q * nthr] = v[q];
  }
  for (; i < n16; i += nthr) dst[i] = src[i];
  if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// bf16x3 pre-split of a weight matrix W [R, K] (fp32, row stride K): row r of `out` has
// split_ld(K) float units; per 16-k chunk 8 dwords of hi pairs then 8 dwords of lo pairs
// (k beyond K zero-filled), see nt_phase_bf3.
__global__ __launch_bounds__(kThreads) void split_bf16x3_kernel(const float* __restrict__ W,
                                                                uint32_t* __restrict__ out, int R,
                                                                int K) {
  const int64_t ld = split_ld(K);
  const int64_t pairs = ld / 2;  // one thread per (row, k pair)
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (idx >= static_cast<int64_t>(R) * pairs) return;
  const int r = static_cast<int>(idx / pairs);
  const int pp = static_cast<int>(idx % pairs);
  const int c = pp / 8, q = pp % 8, k = c * 16 + 2 * q;
  const float x0 = (k < K) ? W[static_cast<int64_t>(r) * K + k] : 0.f;
  const float x1 = (k + 1 < K) ? W[static_cast<int64_t>(r) * K + k + 1] : 0.f;
  const uint32_t hi = pack_bf16(x0, x1);
  const float f0 = __uint_as_float(hi << 16), f1 = __uint_as_float(hi & 0xffff0000u);
  const uint32_t lo = pack_bf16(x0 - f0, x1 - f1);
  uint32_t* o = out + static_cast<int64_t>(r) * ld + c * 16;
  o[q] = hi;
  o[8 + q] = lo;
}

static void launch_split(const float* W, float* out, int R, int K, hipStream_t st) {
  const int64_t n = static_cast<int64_t>(R) * (split_ld(K) / 2);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, W, reinterpret_cast<uint32_t*>(out), R, K);
}

// bf16x3 pre-split of the INPUT rows of the packed steps [0, rows): row p of `out` (split_ld(I)
// float units, same chunk layout as split_bf16x3_kernel) = split(x row of packed row p), the token
// lookup included.  One workgroup per packed row; one pass over the inputs at HBM speed.
struct SplitRowsParams {
  const uint64_t* x_rows;
  const uint64_t* tok_rows;
  const float* emb;
  const int32_t* step_off;
  uint32_t* out;
  int32_t I, vocab, x_step, Tmax;
};

__global__ __launch_bounds__(kThreads) void split_rows_kernel(const SplitRowsParams p) {
  const int64_t pr = blockIdx.x;
  __shared__ rowaddr_t s_src;
  if (threadIdx.x == 0) {
    int lo = 0, hi = p.Tmax - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (static_cast<int64_t>(p.step_off[mid]) <= pr) lo = mid; else hi = mid - 1;
    }
    const int64_t sidx = pr - p.step_off[lo];
    if (p.tok_rows != nullptr) {
      long long tok = reinterpret_cast<const long long*>(p.tok_rows[sidx])[lo];
      tok = tok < 0 ? 0 : (tok >= p.vocab ? p.vocab - 1 : tok);
      s_src = row_addr(p.emb + tok * p.I);
    } else {
      s_src = p.x_rows[sidx] + static_cast<rowaddr_t>(lo) * p.x_step * 4u;
    }
  }
  __syncthreads();
  const float* src = reinterpret_cast<const float*>(s_src);
  const int64_t ld = split_ld(p.I);
  uint32_t* o = p.out + pr * ld;
  for (int pp = threadIdx.x; pp < ld / 2; pp += kThreads) {
    const int c = pp / 8, q = pp % 8, k = c * 16 + 2 * q;
    const float x0 = (k < p.I) ? src[k] : 0.f;
    const float x1 = (k + 1 < p.I) ? src[k + 1] : 0.f;
    const uint32_t hi = pack_bf16(x0, x1);
    const float f0 = __uint_as_float(hi << 16), f1 = __uint_as_float(hi & 0xffff0000u);
    o[c * 16 + q] = hi;
    o[c * 16 + 8 + q] = pack_bf16(x0 - f0, x1 - f1);
  }
}

// small-batch / tiled crossover of the forward steps (Tunables::tiny_max_seqs)
static int tiny_max_seqs() { return tunables().tiny_max_seqs.load(std::memory_order_relaxed); }

// ---------------------------------------------------------------------------------------------
// collate_fn's padding (activity_net/data.py:114-150) as an index kernel: S ragged sequences stored
// back to back (row r of sequence s at src + (first_row[s] + r) * row_bytes) -> the zero-padded
// [S, Tmax, row] block.  One 16-byte (or 4-byte) word per thread, grid-stride; HBM-bound.
// ---------------------------------------------------------------------------------------------
struct PadRowsParams {
  const char* src;
  const int64_t* first_row;
  const int32_t* lens;
  char* dst;
  int64_t words;       // S * Tmax * words_per_row
  int32_t Tmax, words_per_row;
};

template <typename W>
__global__ __launch_bounds__(kThreads) void pad_rows_kernel(const PadRowsParams q) {
  const W* src = reinterpret_cast<const W*>(q.src);
  W* dst = reinterpret_cast<W*>(q.dst);
  const int64_t per_seq = static_cast<int64_t>(q.Tmax) * q.words_per_row;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(kThreads) + threadIdx.x; i < q.words;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t s = i / per_seq, r = i - s * per_seq;
    const int t = static_cast<int>(r / q.words_per_row);
    W v{};
    if (t < q.lens[s]) v = src[q.first_row[s] * q.words_per_row + r];
    dst[i] = v;
  }
}

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
  // bf16x3 serves the LDS-tiled kernels only (the latency-shaped tiny kernel stays exact fp32)
  job->bf3 = bf3 && job->vec && (job->kind_count[0] > tiny_max_seqs() || pool_mode == CMHSE_POOL_ATTN);
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
      const size_t smem = TileSmem<128, 3 * kGruBU>::kBytes;
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
  // tile height of the projection: 64 rows (128 measured equal on the full split: 285.0 against
  // 285.8 ms per pass)
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
      hipLaunchKernelGGL((attn_energy_kernel<true, 2, true, true>), dim3(g1), dim3(kThreads), att_smem, stream, ep);
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
  // 32 single-wave workgroups, 8 x 16 B in flight per lane: 57 GB/s alone (the PCIe Gen5 x16 rate),
  // and the smallest footprint that does it — beside the step kernel every resident pull wave
  // takes a SIMD's free registers, i.e. one of that CU's three step-workgroup slots (sweep in
  // profiles/r02_upload_pipeline.txt: 16 x 256 threads 358 ms / pass, 128 x 64 422, 32 x 64 330)
  constexpr int cap = 32, thr = 64;
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
  unsigned grid = static_cast<unsigned>(workgroups ? workgroups : 8);
  const unsigned thr = 64u * static_cast<unsigned>(waves ? waves : 1);
  const size_t need = (n16 + thr - 1) / thr;
  if (need < grid) grid = static_cast<unsigned>(need ? need : 1);
  hipLaunchKernelGGL(push_bytes_kernel, dim3(grid), dim3(thr), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const float4*>(src), static_cast<float4*>(dst_pinned), n16,
                     static_cast<const unsigned char*>(src) + (n16 << 4),
                     static_cast<unsigned char*>(dst_pinned) + (n16 << 4), tail);
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
