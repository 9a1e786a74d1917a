// gru_step_tile.hpp
//
// The LDS-tiled GRU step: one tile of (64 or 128 sequences) x 64 hidden units x {r, z, n} of one time step
// (gru_step_tile), and its one-launch-per-step kernel.  Replaces nn.GRU over pack_padded_sequence,
// /root/reference/layers.py:97-103.  Included by gru.hip only.
#pragma once

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
  if constexpr (BF3) {
    // pre-split A operands: xs, then hs_s of the previous step (or the pre-split initial states)
    nt_phase_bf3_ring<BM, BNR, MSUB, 3, 4, 2>(smem, ax, bx, I, a_row0, b_row0, acc);
    if (have_h) nt_phase_bf3_ring<BM, BNR, MSUB, 3, 4, 3>(smem, ah, bh, H, a_row0, b_row0, acc);
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

}  // namespace cmhse
