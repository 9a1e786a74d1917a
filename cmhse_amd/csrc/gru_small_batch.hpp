// gru_small_batch.hpp
//
// Small-batch GRU steps (training batches, the ragged few-sequence tails): gru_step_tiny_kernel,
// gru_step_mid_kernel + xproj_kernel (hoisted input projection), gru_fwd_tail_kernel (a training
// chain's <= 32-sequence tail as one resident kernel).  Included by gru.hip only.
#pragma once

namespace cmhse {

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

}  // namespace cmhse
