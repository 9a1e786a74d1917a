// bwd_step_kernels.hpp
//
// One BPTT time step of the packed GRU: the LDS-free step kernels, the mid-size split-K shapes, the <= 32-sequence
// tail as one resident kernel, and the two-launch training-size step (bwd_rec_part_kernel + bwd_gates_kernel).
// Included by bwd.hip only.
#pragma once

namespace cmhse {

// ---------------------------------------------------------------------------------------------
// BPTT step (latency-shaped like gru_step_tiny_kernel): 32 sequences x 32 hidden units per
// workgroup, the 4 waves split K = 3H of  rec = dGh_{t+1} . W_hh  (W_hh^T rows are K-contiguous),
// fixed-order LDS combine, then the gate derivatives of step t in the epilogue.
// ---------------------------------------------------------------------------------------------
struct BwdStepParams {
  const float* dgh_next;  // rows of step t+1: [S_next, 3H]
  const float* whh_t;     // [H, 3H]
  const float* dpool;     // [sumT, H]
  const float* gates;     // [sumT, 4H]
  const float* hs;        // [sumT, H]
  const uint64_t* h0_rows;
  const int32_t* out_row;
  float* carry;  // [S, H]  dh_{t+1} * z_{t+1}
  float* dgx;    // [sumT, 3H]
  float* dgh;    // [sumT, 3H]
  float* dh0;    // [S, H] by out_row (final launch only)
  int32_t H, t, S_t, S_next;
  int64_t off_cur, off_prev;
};

// Up to CMHSE_MAX_JOBS independent BPTT chains share one launch per step (cmhse_gru_pool_bwd_multi):
// workgroups [start[k], start[k+1]) belong to job k, like GruStepGroup in the forward pass.
struct BwdStepGroup {
  BwdStepParams j[CMHSE_MAX_JOBS];
  uint32_t start[CMHSE_MAX_JOBS];
  int32_t n;
};

template <bool VEC, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gru_bwd_step_kernel(const BwdStepGroup grp) {
  CHAIN_WAVE_PRIORITY();
  __shared__ float red[NW][16][64];
  int ji = 0;
#pragma unroll
  for (int k = 1; k < CMHSE_MAX_JOBS; ++k)
    if (k < grp.n && blockIdx.x >= grp.start[k]) ji = k;
  const BwdStepParams& q = grp.j[ji];
  const unsigned wg = blockIdx.x - grp.start[ji];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = q.H, K = 3 * H;
  const int u_tiles = (H + 31) / 32;
  const int u0 = (wg % u_tiles) * 32, m0 = (wg / u_tiles) * 32;
  const int row = lane & 31, hi = lane >> 5;
  f32x16 acc = zero16();
  if (q.S_next > 0) {
    const int m = m0 + row;
    const int mc = (m < q.S_next) ? m : (q.S_next - 1);
    const int u = u0 + row;
    const int uc = (u < H) ? u : (H - 1);
    tiny_phase<VEC, NW>(row_addr(q.dgh_next + static_cast<int64_t>(mc) * K),
                    row_addr(q.whh_t + static_cast<int64_t>(uc) * K), u < H, K, wave, hi, acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // 4 elements per thread: tile row er, columns ec..ec+3
  const int er = tid >> 3, ec = (tid & 7) * 4;
  const int m = m0 + er;
  if (tid >= 256 || m >= q.S_t) return;   // (with NW = 8 the upper four waves only split K)
  const int reg = (er & 3) | ((er >> 3) << 2);
  const int lb = 32 * ((er >> 2) & 1) + ec;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = u0 + ec + j;
    if (u >= H) continue;
    float rec = 0.f;
    if (m < q.S_next)
    {
      rec = q.carry[static_cast<int64_t>(m) * H + u];
      float part = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) part += red[w][reg][lb + j];
      rec += part;
    }
    if (q.t < 0) {  // final launch: d loss / d h0
      q.dh0[static_cast<int64_t>(q.out_row[m]) * H + u] = rec;
      continue;
    }
    const int64_t p = q.off_cur + m;
    const float dh = rec + q.dpool[p * H + u];
    const float* gp = q.gates + p * 4 * H + u;
    const float rg = gp[0], zg = gp[H], ng = gp[2 * H], ghn = gp[3 * H];
    float hp = 0.f;
    if (q.t > 0)
      hp = q.hs[(q.off_prev + m) * H + u];
    else if (q.h0_rows != nullptr)
      hp = reinterpret_cast<const float*>(q.h0_rows[m])[u];
    const float dn_pre = dh * (1.0f - zg) * (1.0f - ng * ng);
    const float dz_pre = dh * (hp - ng) * zg * (1.0f - zg);
    const float dr_pre = dn_pre * ghn * rg * (1.0f - rg);
    float* gx = q.dgx + p * K + u;
    float* gh = q.dgh + p * K + u;
    gx[0] = dr_pre;
    gx[H] = dz_pre;
    gx[2 * H] = dn_pre;
    gh[0] = dr_pre;
    gh[H] = dz_pre;
    gh[2 * H] = dn_pre * rg;
    q.carry[static_cast<int64_t>(m) * H + u] = dh * zg;
  }
}

// ---------------------------------------------------------------------------------------------
// BPTT step for few active sequences (every step of a training batch): the same product on the
// tile shape of the forward mid-size step (gru_step_mid_kernel): 32 (or 16) sequences x BU = 16, 8
// or 4 hidden units per workgroup, one 16 x 16 x 4 MFMA column block, 8 waves split K = 3H with one
// 128-byte line pair in flight each (mid_phase), fixed-order LDS combine, the epilogue's operands
// requested before the K loop.  The 32 x 32 tiles above leave H/32 = 32 workgroups at S_t <= 32,
// each pulling 768 KB of operands through one CU; here 64-256 workgroups pull 430-580 KB each.
// ---------------------------------------------------------------------------------------------
constexpr int kBwdMidNW = 8, kBwdMidRing = 2;

template <int MB, int BU>
__global__ __launch_bounds__(64 * kBwdMidNW) void gru_bwd_step_mid_kernel(const BwdStepGroup grp) {
  CHAIN_WAVE_PRIORITY();
  constexpr int NW = kBwdMidNW, BM = 16 * MB;
  constexpr int OUTS = BM * BU, NOUT = (OUTS + 64 * NW - 1) / (64 * NW);
  __shared__ f32x4v red[NW][MB][64];
  int ji = 0;
#pragma unroll
  for (int k = 1; k < CMHSE_MAX_JOBS; ++k)
    if (k < grp.n && blockIdx.x >= grp.start[k]) ji = k;
  const BwdStepParams& q = grp.j[ji];
  const unsigned wg = blockIdx.x - grp.start[ji];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = q.H, K = 3 * H;
  const int u_tiles = (H + BU - 1) / BU;
  const int u0 = (wg % u_tiles) * BU, m0 = (wg / u_tiles) * BM;
  const int r16 = lane & 15, kq = lane >> 4;
  const bool final_launch = q.t < 0;

  // epilogue operands first (clamped, branch-free): output o = tid + 512 q -> row o / BU, unit o % BU
  float e_carry[NOUT], e_dpool[NOUT], e_g[NOUT][4], e_hp[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int o = tid + 64 * NW * i;
    const int m = m0 + (o / BU) % BM, u = u0 + (o % BU);
    const int mc = (m < q.S_t) ? m : (q.S_t - 1), uc = (u < H) ? u : (H - 1);
    const int mn = (mc < q.S_next) ? mc : 0;
    e_carry[i] = (q.S_next > 0) ? q.carry[static_cast<int64_t>(mn) * H + uc] : 0.f;
    e_dpool[i] = 0.f;
    e_hp[i] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) e_g[i][g] = 0.f;
    if (!final_launch) {
      const int64_t p = q.off_cur + mc;
      e_dpool[i] = q.dpool[p * H + uc];
      const float* gp = q.gates + p * 4 * H + uc;
#pragma unroll
      for (int g = 0; g < 4; ++g) e_g[i][g] = gp[static_cast<int64_t>(g) * H];
      if (q.t > 0)
        e_hp[i] = q.hs[(q.off_prev + mc) * H + uc];
      else if (q.h0_rows != nullptr)
        e_hp[i] = reinterpret_cast<const float*>(q.h0_rows[mc])[uc];
    }
  }

  const bool have_rec = m0 < q.S_next;   // (workgroup-uniform) some row of this block continues
  if (have_rec) {
    rowaddr_t arow[MB], brow[1];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = m0 + 16 * mb + r16;
      const int mc = (m < q.S_next) ? m : (q.S_next - 1);   // rows past S_next: computed, never used
      arow[mb] = row_addr(q.dgh_next + static_cast<int64_t>(mc) * K);
    }
    const int uu = u0 + ((r16 < BU) ? r16 : (BU - 1)), uc = (uu < H) ? uu : (H - 1);
    brow[0] = row_addr(q.whh_t + static_cast<int64_t>(uc) * K);
    f32x4v acc[MB][1];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb][0] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mid_phase<MB, 1, NW, kBwdMidRing>(arow, brow, K, wave, kq, acc);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) red[wave][mb][lane] = acc[mb][0];
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int o = tid + 64 * NW * i;
    if (o >= OUTS) continue;
    const int er = o / BU, eu = o % BU;
    const int m = m0 + er, u = u0 + eu;
    if (m >= q.S_t || u >= H) continue;
    float rec = 0.f;
    if (m < q.S_next) {
      const int mb = er >> 4, rr = er & 15;
      const int sl = (rr >> 2) * 16 + eu, reg = rr & 3;
      float part = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) part += reinterpret_cast<const float*>(&red[w][mb][sl])[reg];
      rec = e_carry[i] + part;
    }
    if (final_launch) {  // d loss / d h0
      q.dh0[static_cast<int64_t>(q.out_row[m]) * H + u] = rec;
      continue;
    }
    const int64_t p = q.off_cur + m;
    const float dh = rec + e_dpool[i];
    const float rg = e_g[i][0], zg = e_g[i][1], ng = e_g[i][2], ghn = e_g[i][3];
    const float hp = e_hp[i];
    const float dn_pre = dh * (1.0f - zg) * (1.0f - ng * ng);
    const float dz_pre = dh * (hp - ng) * zg * (1.0f - zg);
    const float dr_pre = dn_pre * ghn * rg * (1.0f - rg);
    float* gx = q.dgx + p * K + u;
    float* gh = q.dgh + p * K + u;
    gx[0] = dr_pre;
    gx[H] = dz_pre;
    gx[2 * H] = dn_pre;
    gh[0] = dr_pre;
    gh[H] = dz_pre;
    gh[2 * H] = dn_pre * rg;
    q.carry[static_cast<int64_t>(m) * H + u] = dh * zg;
  }
}

// ---------------------------------------------------------------------------------------------
// The long few-sequence TAIL of a BPTT chain as ONE resident kernel.  The whole-paragraph /
// whole-video sequences of a training batch run tens of steps past the last sentence / clip with
// at most 32 sequences still active (ActivityNet, batch 32: ~95 of the text chain's 124 steps,
// and every step of the level-2 encoders and of the decoders);
// backward those steps come FIRST, each a dependent launch of gru_bwd_step_mid_kernel<1, 16> —
// 12.8 us apiece on an idle chip, 24 us beside the other tower's chain — and the step's own work
// is a 16 x 16 output tile per workgroup.  Here the H / 16 workgroups of that kernel stay resident
// from step Tmax - 1 down to the first step with more than 16 active sequences:
//   * the workgroup's W_hh^T slice (16 columns x 3H) is loaded into registers ONCE, in the MFMA
//     operand layout of mid_phase;
//   * per step, the rows dGh_{t+1} — written one step earlier by ALL workgroups — are the only
//     operand that crosses workgroups.  The 8 XCDs' L2s are not coherent with each other, and the
//     cache maintenance the HIP memory model prescribes for that (write-back + invalidate per
//     fence) costs 24-74 us per step (tools/microbench/grid_barrier.hip).  But every dGh row is
//     written exactly once, to an address nobody read before, and read only after the step's
//     barrier: agent-scope (sc1) stores that write through and sc1 loads that bypass the
//     non-coherent caches are enough — 1.8-4.2 us for the barrier itself (64 / 256 workgroups), no
//     cache maintenance at all;
//   * the barrier is a counter in the call's workspace: one agent-scope atomic add per workgroup
//     and a bounded spin.  All workgroups are co-resident by construction (at most 256 of them,
//     8 waves and 8 KB of LDS each; nothing they wait for waits for them); should they not be (a
//     shared GPU), the barrier times out and the call is reported failed (grid_sync.hpp);
//   * carry (dh_{t+1} z_{t+1}) lives in a register of the thread that owns the output.
// Block ownership of the 8 waves, accumulation order and combine order are those of
// gru_bwd_step_mid_kernel<1, 16>; the results agree with it to fp32 rounding (the compiler
// contracts the gate arithmetic of the two kernels into different FMAs) and are bitwise
// reproducible from run to run (tested).
// ---------------------------------------------------------------------------------------------
struct BwdTailParams {
  const float* whh_t;       // [H, 3H]
  const float* dpool;       // [sumT, H]
  const float* gates;       // [sumT, 4H]
  const float* hs;          // [sumT, H]
  const int32_t* step_off;  // device [Tmax + 1]
  float* carry;             // [S, H]: written for the rows of step t_lo when the kernel ends
  float* dgx;               // [sumT, 3H]
  float* dgh;               // [sumT, 3H]
  GridSync sync;            // grid barrier words (zeroed by the caller)
  int32_t H, t_hi, t_lo;    // steps t_hi = Tmax - 1 down to t_lo >= 1, at most 32 active sequences each
};

constexpr int kTailMaxSeqs = 32;   // two 16-row blocks per workgroup

template <int KBMAX, int MB>
__global__ __launch_bounds__(512) void gru_bwd_tail_kernel(const BwdTailParams q) {
  CHAIN_WAVE_PRIORITY();
  constexpr int NW = 8;
  __shared__ f32x4v red[NW][MB][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = q.H, K = 3 * H;
  const int u0 = blockIdx.x * 16;
  const int r16 = lane & 15, kq = lane >> 4;
  const int nkb = K / 16;
  auto block_of = [&](int i) { return (i >> 1) * 2 * NW + 2 * wave + (i & 1); };   // mid_phase's ownership
  int nmine = 0;
  while (nmine < KBMAX && block_of(nmine) < nkb) ++nmine;
  // B operand: column r16 of the tile = unit u0 + r16 of W_hh^T, resident for the whole tail
  float4 wreg[KBMAX];
  {
    const int uu = u0 + r16, uc = (uu < H) ? uu : (H - 1);
    const float* brow = q.whh_t + static_cast<int64_t>(uc) * K;
#pragma unroll
    for (int i = 0; i < KBMAX; ++i)
      wreg[i] = (i < nmine) ? *reinterpret_cast<const float4*>(brow + block_of(i) * 16 + 4 * kq)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the output this thread owns (threads 0..255): tile row er = sequence, unit u
  const int er = tid >> 4, eu = tid & 15;
  const int u = u0 + eu;
  const bool owner = er < 16 * MB && u < H;     // threads 0..255 at MB = 1, all 512 at MB = 2
  float carry = 0.f;
  unsigned arrivals = 0;
  int S_next = 0;
  for (int t = q.t_hi; t >= q.t_lo; --t) {
    const int off_cur = q.step_off[t], off_next = q.step_off[t + 1], off_prev = q.step_off[t - 1];
    const int S_t = off_next - off_cur;
    // the epilogue's own operands do not depend on the chain: request them before the product
    float e_dpool = 0.f, e_g[4] = {0.f, 0.f, 0.f, 0.f}, e_hp = 0.f;
    const int64_t p = off_cur + er;
    if (owner && er < S_t) {
      e_dpool = q.dpool[p * H + u];
      const float* gp = q.gates + p * 4 * H + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) e_g[g] = gp[static_cast<int64_t>(g) * H];
      e_hp = q.hs[(static_cast<int64_t>(off_prev) + er) * H + u];
    }
    if (S_next > 0) {
      // A operand: rows of step t + 1, published by every workgroup before the barrier below;
      // one 16-row block at a time (its 96 registers are reused by the second block)
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(q.dgh + static_cast<int64_t>(off_next) * K), 0, 0x7fffffff, 0x00020000);
      typedef int i32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        if (16 * mb < S_next) {      // (workgroup-uniform)
          const int m = 16 * mb + r16;
          const int row_b = ((m < S_next) ? m : (S_next - 1)) * K * 4;
          i32x4v areg[KBMAX];
#pragma unroll
          for (int i = 0; i < KBMAX; ++i)
            if (i < nmine)
              areg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, row_b + (block_of(i) * 16 + 4 * kq) * 4, 0, 16);
#pragma unroll
          for (int i = 0; i < KBMAX; ++i) {
            if (i >= nmine) continue;   // wave-uniform
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(areg[i].x), wreg[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(areg[i].y), wreg[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(areg[i].z), wreg[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(areg[i].w), wreg[i].w, acc, 0, 0, 0);
          }
        }
        red[wave][mb][lane] = acc;
      }
      __syncthreads();
    }
    if (owner && er < S_t) {
      float rec = 0.f;
      if (er < S_next) {
        const int mb = er >> 4, rr = er & 15;
        const int sl = (rr >> 2) * 16 + eu, reg = rr & 3;
        float part = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) part += reinterpret_cast<const float*>(&red[w][mb][sl])[reg];
        rec = carry + part;
      }
      const float dh = rec + e_dpool;
      const float rg = e_g[0], zg = e_g[1], ng = e_g[2], ghn = e_g[3];
      const float dn_pre = dh * (1.0f - zg) * (1.0f - ng * ng);
      const float dz_pre = dh * (e_hp - ng) * zg * (1.0f - zg);
      const float dr_pre = dn_pre * ghn * rg * (1.0f - rg);
      float* gx = q.dgx + p * K + u;
      float* gh = q.dgh + p * K + u;
      gx[0] = dr_pre;
      gx[H] = dz_pre;
      gx[2 * H] = dn_pre;
      // the next step's A operand, in every workgroup: write through to where all XCDs see it
      __hip_atomic_store(gh, dr_pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gh + H, dz_pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gh + 2 * H, dn_pre * rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      carry = dh * zg;
    }
    S_next = S_t;
    if (t == q.t_lo) break;
    // grid barrier: this workgroup's rows are written through, then everybody's are
    __builtin_amdgcn_s_waitcnt(0);
    arrivals += gridDim.x;
    if (!grid_sync_wait(q.sync, arrivals)) return;
  }
  if (owner && er < S_next) q.carry[static_cast<int64_t>(er) * H + u] = carry;
}

// ---------------------------------------------------------------------------------------------
// BPTT step of a training-size batch (32 < S_t <= bwd_mid_max_seqs) as TWO launches that move a
// third of the bytes.  The product of a step,  rec[S, H] = dGh_{t+1}[S, 3H] . W_hh[3H, H],  is a
// skinny GEMM: a handful of row tiles, K = 3H.  gru_bwd_step_mid_kernel covers it with 32 x 16
// tiles that each walk ALL of K — 320 workgroups x 576 KB = 184 MB of operands through the cache
// fabric per step at S_t = 152, which is what bounds it (8 TB/s for 1 GFLOP) and what makes two
// chains and the weight-gradient products beside them slow each other down.  Here:
//   bwd_rec_part_kernel   32 x 128 tiles, K cut into `splits` slices over the grid's second
//                         dimension (240-256 workgroups in all): operands staged through LDS in
//                         32-k chunks (whole 128-byte lines per row), eight waves each owning 16
//                         columns (two 16x16x4 accumulators); writes the slice's partial tile to scratch.  Operand bytes per step:
//                         outputs x K x 4 x (1/128 + 1/32) = 76 MB at S_t = 152.
//   bwd_gates_kernel      adds the partials in slice order (bitwise reproducible), then the gate
//                         derivatives of step t exactly as the one-launch kernels' epilogue.
// (Measured late in round 3: both in ONE launch — a ticket per tile, the last K slice to arrive adds
// the slices and evaluates the tile's gates, partials exchanged through agent-scope stores / loads —
// is correct and 0.1 ms per training step SLOWER: the epilogue then waits for the slowest slice and
// runs on 32 workgroups instead of the whole chip; the second launch's gap is cheaper than that.)
// ---------------------------------------------------------------------------------------------
constexpr int kRecBM = 32, kRecBN = 128, kRecBK = 32, kRecLd = kRecBK + 4;

struct RecPartParams {
  const float* a;     // dGh_{t+1} rows [S_next, K]
  const float* b;     // W_hh^T rows [H, K]
  float* part;        // [splits][m_pad][H]
  int32_t S_next, H, K, k_slice, m_pad, n_tiles;
};

constexpr int kRecThreads = 512;   // 8 waves: two per SIMD, so one wave's chunk barrier and LDS round trip hide under the other's MFMAs

// The product of one (32-row block, 128-column tile, K slice): acc[mb][reg] = element (row 16 mb +
// 4 kq + reg, column 16 wave + r16) of the slice's partial tile.  Ends with a workgroup barrier
// (the LDS buffers may be reused).
typedef float (*RecLds)[(kRecBM + kRecBN) * kRecLd];
__device__ __forceinline__ void rec_part_tile(const RecPartParams& q, RecLds lds, const int m0, const int n0,
                                              const int y, f32x4v (&acc)[2]) {
  constexpr int ROWS = kRecBM + kRecBN, PIECES = ROWS * (kRecBK / 4), NP = (PIECES + kRecThreads - 1) / kRecThreads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = y * q.k_slice;
  const int k1 = (k0 + q.k_slice < q.K) ? (k0 + q.k_slice) : q.K;
  const int nchunks = (k1 - k0 + kRecBK - 1) / kRecBK;
  // staging: a row of a chunk is 32 floats = 8 x 16 B: piece p = tid + 512 i -> staged row p >> 3
  // (0..31 rows of dGh, 32..159 rows of W_hh^T), slot p & 7 — eight lanes read one whole 128-byte line
  rowaddr_t rbase[NP];
  int lds_off[NP], slot_k[NP];
  bool live[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = tid + kRecThreads * i;
    live[i] = pc < PIECES;
    const int row = live[i] ? (pc >> 3) : 0, slot = pc & 7;
    if (row < kRecBM) {
      const int m = m0 + row;
      rbase[i] = row_addr(q.a + static_cast<int64_t>(m < q.S_next ? m : (q.S_next - 1)) * q.K);
    } else {
      const int n = n0 + row - kRecBM;
      rbase[i] = row_addr(q.b + static_cast<int64_t>(n < q.H ? n : (q.H - 1)) * q.K);
    }
    slot_k[i] = slot * 4;
    lds_off[i] = row * kRecLd + slot * 4;
  }
  float4 r[NP];
  auto issue = [&](int c) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (live[i]) r[i] = issue_row4<true>(rbase[i], k0 + c * kRecBK + slot_k[i], k1);
  };
  auto stage = [&](int buf, int c) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (live[i])
        *reinterpret_cast<float4*>(&lds[buf][lds_off[i]]) =
            finish_row4<true>(r[i], true, k0 + c * kRecBK + slot_k[i], k1);
  };
  // wave w owns the 16 columns 16 w .. 16 w + 15 of the 32 x 128 tile, both 16-row blocks:
  // v_mfma_f32_16x16x4_f32, lane (r16 = lane & 15, kq = lane >> 4) feeds k = 4 kq + j of a 16-k block
  acc[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, kq = lane >> 4;
  auto compute = [&](int cur) {
    const float* A = &lds[cur][0] + r16 * kRecLd + 4 * kq;
    const float* B = &lds[cur][0] + (kRecBM + 16 * wave + r16) * kRecLd + 4 * kq;
#pragma unroll
    for (int kb = 0; kb < kRecBK / 16; ++kb) {
      const float4 a0 = *reinterpret_cast<const float4*>(A + kb * 16);
      const float4 a1 = *reinterpret_cast<const float4*>(A + 16 * kRecLd + kb * 16);
      const float4 b = *reinterpret_cast<const float4*>(B + kb * 16);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc[1], 0, 0, 0);
    }
  };
  issue(0);
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunks) issue(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur);
    if (c + 1 < nchunks) stage(cur ^ 1, c + 1);
    __syncthreads();
  }
}

__global__ __launch_bounds__(kRecThreads) void bwd_rec_part_kernel(const RecPartParams q) {
  CHAIN_WAVE_PRIORITY();
  __shared__ __attribute__((aligned(16))) float lds[2][(kRecBM + kRecBN) * kRecLd];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int n0 = static_cast<int>(blockIdx.x % q.n_tiles) * kRecBN;
  const int m0 = static_cast<int>(blockIdx.x / q.n_tiles) * kRecBM;
  f32x4v acc[2];
  rec_part_tile(q, lds, m0, n0, static_cast<int>(blockIdx.y), acc);
  // the slice's partial tile (element (row r, col c) of a 16 x 16 block: lane (r >> 2) * 16 + c,
  // register r & 3); rows past S_next hold a clamped row's garbage and are never read
  float* P = q.part + (static_cast<int64_t>(blockIdx.y) * q.m_pad + m0) * q.H;
  const int n = n0 + 16 * wave + r16;
  if (n < q.H) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        P[static_cast<int64_t>(16 * mb + 4 * kq + reg) * q.H + n] = acc[mb][reg];
  }
}

struct GatesBwdParams {
  BwdStepParams s;
  const float* part;   // [splits][m_pad][H] or NULL (no continuing gradient: the chain's first launch)
  int32_t splits, m_pad;
};

// one thread per (sequence, 4 hidden units)
__global__ __launch_bounds__(kThreads) void bwd_gates_kernel(const GatesBwdParams g) {
  CHAIN_WAVE_PRIORITY();
  const BwdStepParams& q = g.s;
  const int H = q.H, K = 3 * H, h4 = H / 4;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= static_cast<int64_t>(q.S_t) * h4) return;
  const int m = static_cast<int>(e / h4), u = static_cast<int>(e % h4) * 4;
  float4 rec = zero4();
  if (m < q.S_next) {
    rec = *reinterpret_cast<const float4*>(q.carry + static_cast<int64_t>(m) * H + u);
    float4 sum = zero4();
    for (int y = 0; y < g.splits; ++y) {
      const float4 p = *reinterpret_cast<const float4*>(
          g.part + (static_cast<int64_t>(y) * g.m_pad + m) * H + u);
      sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
    }
    rec.x += sum.x; rec.y += sum.y; rec.z += sum.z; rec.w += sum.w;
  }
  if (q.t < 0) {  // final launch: d loss / d h0
    *reinterpret_cast<float4*>(q.dh0 + static_cast<int64_t>(q.out_row[m]) * H + u) = rec;
    return;
  }
  const int64_t p = q.off_cur + m;
  const float4 dp = *reinterpret_cast<const float4*>(q.dpool + p * H + u);
  const float* gp = q.gates + p * 4 * H + u;
  const float4 rg = *reinterpret_cast<const float4*>(gp);
  const float4 zg = *reinterpret_cast<const float4*>(gp + H);
  const float4 ng = *reinterpret_cast<const float4*>(gp + 2 * H);
  const float4 ghn = *reinterpret_cast<const float4*>(gp + 3 * H);
  float4 hp = zero4();
  if (q.t > 0)
    hp = *reinterpret_cast<const float4*>(q.hs + (q.off_prev + m) * H + u);
  else if (q.h0_rows != nullptr)
    hp = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(q.h0_rows[m]) + u);
  float4 drp, dzp, dnp, dnr, car;
#define GATE_LANE_(c)                                            \
  {                                                                   \
    const float dh = rec.c + dp.c;                                    \
    const float dn_pre = dh * (1.0f - zg.c) * (1.0f - ng.c * ng.c);   \
    dzp.c = dh * (hp.c - ng.c) * zg.c * (1.0f - zg.c);                \
    drp.c = dn_pre * ghn.c * rg.c * (1.0f - rg.c);                    \
    dnp.c = dn_pre;                                                   \
    dnr.c = dn_pre * rg.c;                                            \
    car.c = dh * zg.c;                                                \
  }
  GATE_LANE_(x) GATE_LANE_(y) GATE_LANE_(z) GATE_LANE_(w)
#undef GATE_LANE_
  float* gx = q.dgx + p * K + u;
  float* gh = q.dgh + p * K + u;
  *reinterpret_cast<float4*>(gx) = drp;
  *reinterpret_cast<float4*>(gx + H) = dzp;
  *reinterpret_cast<float4*>(gx + 2 * H) = dnp;
  *reinterpret_cast<float4*>(gh) = drp;
  *reinterpret_cast<float4*>(gh + H) = dzp;
  *reinterpret_cast<float4*>(gh + 2 * H) = dnr;
  *reinterpret_cast<float4*>(q.carry + static_cast<int64_t>(m) * H + u) = car;
}

}  // namespace cmhse
