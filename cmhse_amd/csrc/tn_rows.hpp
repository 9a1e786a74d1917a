// tn_rows.hpp — weight-gradient products straight from the packed rows (exact fp32 MFMA).
//
//   C[m][n] (+)= sum_{p = p0}^{p1 - 1} A[p][m] * B_p[n]          bias[m] (+)= sum_p A[p][m]
//
// is the shape of every weight gradient of the path: dW_ih = dGx^T X, dW_hh = dGh^T H_prev,
// dW_lin = dU^T Hs (and db_ih, db_hh, db_lin, the column sums of the same A).  The contraction
// index is the PACKED ROW p = (time step, sequence): both operands lie row-major with p as the
// row index, so neither is K-contiguous.  Round 2 copied both through HBM into K-major form
// (gather_transpose) and ran the NT loop over all rows once the BPTT chain had finished.  Here the
// transposition happens on the way into LDS:
//   * global -> registers: thread (c = tid & 127, kg = tid >> 7) loads column c of the 8 packed
//     rows kg*8 .. kg*8+7 of a 16-row chunk — one dword per row, a wave reading 256 contiguous
//     bytes of a row; the row base is wave-uniform (a buffer descriptor in scalar registers), the
//     lane offset constant over the loop, so a load costs no vector-ALU instruction
//     (nt_core.hpp: vector-ALU work in the K loop costs matrix time);
//   * registers -> LDS: two ds_write_b128 put those 8 k of column c into the [row = c][16 k] tile
//     layout of nt_core.hpp (row stride 20 floats: conflict-free for these writes too);
//   * the MFMA side is nt_phase's: one ds_read_b128 per operand feeds four v_mfma_f32_32x32x2_f32,
//     the same rotated software pipeline, one barrier per 16 rows.
// Because a row RANGE is an argument, the product can be taken chunk by chunk while the BPTT chain
// is still producing earlier time steps (rows of steps >= t are final once the chain has passed t):
// the launches ride on a side stream beside the latency-bound chain (bwd.hip), accumulate in launch
// order (no atomics: the gradients are bitwise reproducible), and several products over the same
// row range share one launch (TnRowsGroup).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nt_core.hpp"

namespace cmhse {

struct TnRowsProblem {
  const float* a;          // column m of packed row p at a + p * lda + m
  int64_t lda;
  const uint64_t* b_addr;  // [rows] address of the B row of packed row p (N floats)
  float* c;                // [M, ldc]
  int64_t ldc;
  float* bias;             // [M] column sums of A over the row range, or NULL
  int32_t M, N, n_tiles;   // n_tiles = ceil(N / 128)
};

// Tile heights (columns of A = rows of C per workgroup) the kernel is built for.
//   128: each wave 64 x 64 of C, 32 MFMAs per barrier;
//   192: each wave 96 x 64, 48 MFMAs per barrier.  3H = 3072 is a whole number of either.
// Three workgroups per CU both.
constexpr int kTnRowsBmSmall = 128, kTnRowsBmTall = 192;

constexpr int kTnRowsMaxProblems = 4;

struct TnRowsGroup {
  TnRowsProblem q[kTnRowsMaxProblems];
  uint32_t start[kTnRowsMaxProblems];   // first workgroup of problem k
  int32_t n;
  int32_t accumulate;   // 0: C (and bias) = the product over the range; 1: += (a later chunk)
  int64_t p0, p1;       // the packed rows contracted over
  // Row split (gridDim.y = splits > 1): workgroup (x, y) contracts rows [p0 + y * seg, ...) of tile
  // x and stores its part of C (dense, leading dimension N) and of the bias into the scratch block
  // of split y: part + y * part_stride + part_off[problem] (C, then M floats of bias);
  // tn_rows_reduce_kernel then adds the parts to C in the order of y.  More workgroups per tile —
  // a launch with few tiles fills the chip, several share a CU and hide each other's latencies —
  // and the sum is still taken in a fixed order: bitwise reproducible, no atomics.
  int32_t seg;          // rows per split (multiple of 16); 0 = no split
  float* part;
  int64_t part_stride;
  int64_t part_off[kTnRowsMaxProblems];
};

// Three waves per SIMD (<= 168 registers) either way: three workgroups share a CU and hide each
// other's barrier and load latencies (128-row tile: 115-119 TFLOP/s on a launch of exactly three
// tiles per CU against 66 with two; four per CU — 128 registers, 2 spilled — 94).
// MS = 2: BM = 128.  MS = 3: BM = 192, 96 accumulator registers; with the two fragment sets of the
// rotated pipeline it needs 184 registers (two waves per SIMD: 115 TFLOP/s at an exact fill, no
// better than the small tile), with ONE fragment set 166: 122-125 TFLOP/s (K = 9600; I = 5120 /
// 2048, H = 1024; profiles/r04_wgrad_rate.txt) — the barrier is amortised over 48 MFMAs per wave
// instead of 32, the ratio of the validation pass's step kernel.
template <int MS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3)))
void gemm_tn_rows_kernel(const TnRowsGroup g) {
  constexpr int BM = 64 * MS, BN = 128, MSUB = MS, NSUB = 2;
  constexpr bool TALL = (MS == 3);
  using SM = TileSmem<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int qi = 0;
#pragma unroll
  for (int k = 1; k < kTnRowsMaxProblems; ++k)
    if (k < g.n && blockIdx.x >= g.start[k]) qi = k;
  const TnRowsProblem& q = g.q[qi];
  const unsigned wg = blockIdx.x - g.start[qi];
  const int n0 = static_cast<int>(wg % q.n_tiles) * BN, m0 = static_cast<int>(wg / q.n_tiles) * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = tid & 127;
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 7);   // waves 0,1: rows 0..7; waves 2,3: rows 8..15
  // TALL: columns 128..191 of the A tile — thread (c2 = 128 + lane, kq = wave) loads column c2 of
  // the 4 packed rows kq*4 .. kq*4+3 of the chunk
  const int c2 = 128 + lane;
  const int kq = __builtin_amdgcn_readfirstlane(wave);
  // this thread's column of the A tile / B tile, clamped into the matrix (columns past the edge
  // compute on a valid column's data and are never stored)
  const unsigned am = static_cast<unsigned>((m0 + c < q.M) ? (m0 + c) : (q.M - 1));
  const unsigned bn = static_cast<unsigned>((n0 + c < q.N) ? (n0 + c) : (q.N - 1));
  const unsigned am2 = static_cast<unsigned>((m0 + c2 < q.M) ? (m0 + c2) : (q.M - 1));
  int64_t p0 = g.p0, p1 = g.p1;
  if (g.seg > 0) {
    p0 += static_cast<int64_t>(blockIdx.y) * g.seg;
    p1 = (p0 + g.seg < p1) ? (p0 + g.seg) : p1;
  }
  const int nchunks = static_cast<int>((p1 - p0 + kBK - 1) / kBK);   // (>= 1: the launcher sizes seg so)
  const bool want_bias = (q.bias != nullptr) && (n0 == 0);   // workgroup-uniform
  // Row bases are wave-uniform and live in scalar registers.  Loads are BUFFER loads: resource
  // descriptor (SGPRs: the row's base address) + this lane's constant 32-bit byte offset + a scalar
  // offset — no vector-ALU instruction per load.  A rows: one descriptor at the first row of the
  // range, the row selected by the scalar offset; B rows: a descriptor per row, its base read from
  // the address table through SCALAR loads (constant address space), one chunk ahead of the vector
  // loads that use it.  Row indices are relative to p0 and 32-bit (the launcher bounds the range
  // so that row * lda * 4 stays below 2^31).
  typedef const __attribute__((address_space(4))) uint64_t* cptr_u64;
  constexpr int kRsrcFlags = 0x00020000;     // raw buffer, 32-bit data format (gfx9 / CDNA)
  const int nrows = static_cast<int>(p1 - p0);
  const int lda_b = static_cast<int>(q.lda) * 4;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(q.a + p0 * q.lda), 0, 0x7fffffff, kRsrcFlags);
  cptr_u64 const b_tab = (cptr_u64)(reinterpret_cast<uintptr_t>(q.b_addr + p0));
  const unsigned a_off = am * 4u, b_off = bn * 4u, a2_off = am2 * 4u;

  float ra[8], rb[8];
  float ra2[TALL ? 4 : 1];
  float bsum2 = 0.f;
  rowaddr_t nb[8];          // B row bases of the next chunk to load (scalar)
  float bsum = 0.f;
  f32x16 acc[MSUB][NSUB];
#pragma unroll
  for (int i = 0; i < MSUB; ++i)
#pragma unroll
    for (int j = 0; j < NSUB; ++j) acc[i][j] = zero16();

  // row j of this wave's half of chunk ch, relative to p0, clamped into the range (the tail chunk
  // re-reads the last row; write_lds masks it)
  auto row_of = [&](int ch, int j) {
    const int r = ch * kBK + kg * 8 + j;
    return (r < nrows) ? r : (nrows - 1);
  };
  auto row2_of = [&](int ch, int j) {
    const int r = ch * kBK + kq * 4 + j;
    return (r < nrows) ? r : (nrows - 1);
  };
  auto fetch_rows = [&](int ch) {
#pragma unroll
    for (int j = 0; j < 8; ++j) nb[j] = b_tab[row_of(ch, j)];
  };
  // loads of chunk ch: 8 packed rows, one dword of each operand per row (B bases from `nb`)
  auto issue_global = [&](int ch) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, a_off, row_of(ch, j) * lda_b, 0));
      const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(nb[j]), 0, 0x7fffffff, kRsrcFlags);
      rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rs, b_off, 0, 0));
    }
    if (TALL) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ra2[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, a2_off, row2_of(ch, j) * lda_b, 0));
    }
  };
  auto write_lds = [&](int buf, int ch, auto maskedc) {
    constexpr bool MASKED = decltype(maskedc)::value;
    if (MASKED) {
      const int rk = ch * kBK + kg * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (rk + j >= nrows) ra[j] = 0.f;    // A = 0 removes the row from C and from the bias
      if (TALL) {
        const int rq = ch * kBK + kq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (rq + j >= nrows) ra2[j] = 0.f;
      }
    }
    if (want_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += ra[j];
      if (TALL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum2 += ra2[j];
      }
    }
    float* ap = SM::a(smem, buf) + c * kLdsLd + kg * 8;
    float* bp = SM::b(smem, buf) + c * kLdsLd + kg * 8;
    *reinterpret_cast<float4*>(ap) = make_float4(ra[0], ra[1], ra[2], ra[3]);
    *reinterpret_cast<float4*>(ap + 4) = make_float4(ra[4], ra[5], ra[6], ra[7]);
    *reinterpret_cast<float4*>(bp) = make_float4(rb[0], rb[1], rb[2], rb[3]);
    *reinterpret_cast<float4*>(bp + 4) = make_float4(rb[4], rb[5], rb[6], rb[7]);
    if (TALL)
      *reinterpret_cast<float4*>(SM::a(smem, buf) + c2 * kLdsLd + kq * 4) =
          make_float4(ra2[0], ra2[1], ra2[2], ra2[3]);
  };
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int a_row0 = wm * (BM / 2);
  const int b_row0[NSUB] = {wn * 64, wn * 64 + 32};
  float4 f0a[MSUB], f0b[NSUB], f1a[MSUB], f1b[NSUB];
  auto read_frags = [&](int buf, int kb, float4(&fa)[MSUB], float4(&fb)[NSUB]) {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
      fa[ms] = *reinterpret_cast<const float4*>(SM::a(smem, buf) +
                                                (a_row0 + ms * 32 + frow) * kLdsLd + kb * 8 + fk);
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
      fb[ns] = *reinterpret_cast<const float4*>(SM::b(smem, buf) +
                                                (b_row0[ns] + frow) * kLdsLd + kb * 8 + fk);
  };
  auto mfma_block = [&](const float4(&fa)[MSUB], const float4(&fb)[NSUB], auto j0c, auto j1c) {
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
#pragma unroll
      for (int ms = 0; ms < MSUB; ++ms) {
        const float av = (j == 0) ? fa[ms].x : (j == 1) ? fa[ms].y : (j == 2) ? fa[ms].z : fa[ms].w;
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          const float bv = (j == 0) ? fb[ns].x : (j == 1) ? fb[ns].y : (j == 2) ? fb[ns].z : fb[ns].w;
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ms][ns], 0, 0, 0);
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using Plain = std::integral_constant<bool, false>;
  using Masked = std::integral_constant<bool, true>;
  // chunk ch lies wholly inside the range?
  auto full = [&](int ch) { return (ch + 1) * kBK <= nrows; };

  if constexpr (TALL) {
    // One fragment set (the second would not fit three waves per SIMD beside 96 accumulators): the
    // fragments of the chunk's second half are requested once the MFMAs of its first half have
    // issued; the other two workgroups of the CU cover that latency.  Same k order as below.
    using I0_ = std::integral_constant<int, 0>;
    using I4_ = std::integral_constant<int, 4>;
    fetch_rows(0);
    issue_global(0);
    if (nchunks > 1) fetch_rows(1);
    if (full(0)) write_lds(0, 0, Plain{}); else write_lds(0, 0, Masked{});
    for (int ch = 0; ch < nchunks; ++ch) {
      const int cur = ch & 1;
      const bool more = ch + 1 < nchunks;      // uniform
      __syncthreads();
      read_frags(cur, 0, f0a, f0b);
      if (more) {
        issue_global(ch + 1);
        if (ch + 2 < nchunks) fetch_rows(ch + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I0_{}, I4_{});
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 1, f0a, f0b);
      if (more) {
        if (full(ch + 1)) write_lds(cur ^ 1, ch + 1, Plain{}); else write_lds(cur ^ 1, ch + 1, Masked{});
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I0_{}, I4_{});
    }
    __syncthreads();
  } else {
    // the rotated software pipeline of nt_phase (see there): only the barrier is exposed
    fetch_rows(0);
    issue_global(0);
    if (nchunks > 1) fetch_rows(1);
    if (full(0)) write_lds(0, 0, Plain{}); else write_lds(0, 0, Masked{});
    __syncthreads();
    read_frags(0, 0, f0a, f0b);
    if (nchunks > 1) {
      issue_global(1);
      if (nchunks > 2) fetch_rows(2);
    }
    read_frags(0, 1, f1a, f1b);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f0a, f0b, I0{}, I3{});
    __builtin_amdgcn_sched_barrier(0);
    if (nchunks > 1) {
      if (full(1)) write_lds(1, 1, Plain{}); else write_lds(1, 1, Masked{});
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f0a, f0b, I3{}, I4{});
    for (int ch = 1; ch < nchunks; ++ch) {
      const int cur = ch & 1;
      const bool more = ch + 1 < nchunks;      // uniform
      __syncthreads();
      read_frags(cur, 0, f0a, f0b);
      if (more) {
        issue_global(ch + 1);
        if (ch + 2 < nchunks) fetch_rows(ch + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f1a, f1b, I0{}, I4{});
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 1, f1a, f1b);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I0{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if (full(ch + 1)) write_lds(cur ^ 1, ch + 1, Plain{}); else write_lds(cur ^ 1, ch + 1, Masked{});
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I3{}, I4{});
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f1a, f1b, I0{}, I4{});
    __syncthreads();
  }

  // ---- epilogue ----
  const bool accum = g.accumulate != 0 && g.seg == 0;    // (a split's part is always stored)
  float* cbase = q.c;
  float* bias = q.bias;
  int64_t ldc = q.ldc;
  if (g.seg > 0) {
    cbase = g.part + static_cast<int64_t>(blockIdx.y) * g.part_stride + g.part_off[qi];
    bias = cbase + static_cast<int64_t>(q.M) * q.N;
    ldc = q.N;
  }
  if (want_bias) {
    // column sums: the two row halves (kg) of every column meet in LDS, fixed order
    float* red = smem;
    if (kg == 1) red[c] = bsum;
    __syncthreads();
    if (kg == 0 && m0 + c < q.M) {
      const float s = bsum + red[c];
      float* dst = bias + m0 + c;
      *dst = accum ? (*dst + s) : s;
    }
    if (TALL) {
      // columns 128..191: the four row quarters (kq) of a column, added in the order of kq
      float* red2 = smem + 128;
      if (kq > 0) red2[(kq - 1) * 64 + lane] = bsum2;
      __syncthreads();
      if (kq == 0 && m0 + c2 < q.M) {
        const float s = ((bsum2 + red2[lane]) + red2[64 + lane]) + red2[128 + lane];
        float* dst = bias + m0 + c2;
        *dst = accum ? (*dst + s) : s;
      }
    }
  }
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + a_row0 + ms * 32 + acc_row(r, lane);
      if (m >= q.M) continue;
      float* crow = cbase + static_cast<int64_t>(m) * ldc;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const int n = n0 + b_row0[ns] + acc_col(lane);
        if (n >= q.N) continue;
        const float v = acc[ms][ns][r];
        crow[n] = accum ? (crow[n] + v) : v;
      }
    }
}

// C[m][n] (+)= part_0[m][n] + part_1[m][n] + ... and bias[m] likewise, in split order.
struct TnRowsReduce {
  const float* part;
  int64_t part_stride, part_off;
  float* c;
  int64_t ldc;
  float* bias;
  int32_t M, N, splits, accumulate;
};

__global__ __launch_bounds__(kThreads) void tn_rows_reduce_kernel(const TnRowsReduce q) {
  const int64_t total = static_cast<int64_t>(q.M) * q.N + (q.bias ? q.M : 0);
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * kThreads) {
    float s = 0.f;
    for (int y = 0; y < q.splits; ++y) s += q.part[y * q.part_stride + q.part_off + e];
    float* dst;
    if (e < static_cast<int64_t>(q.M) * q.N) dst = q.c + (e / q.N) * q.ldc + (e % q.N);
    else dst = q.bias + (e - static_cast<int64_t>(q.M) * q.N);
    *dst = q.accumulate ? (*dst + s) : s;
  }
}

}  // namespace cmhse
