// gru_attention.hpp
//
// Attention pooling of layers.Attention.forward (/root/reference/layers.py:105-117): the energy tiles
// e = w_att . tanh(W_lin h + b_lin) and the masked exp-softmax + weighted sum.  Included by gru.hip only.
#pragma once

namespace cmhse {

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
  if constexpr (BF3 && ASPLIT)      // both operands pre-split: the LDS-DMA ring (rows past the edges are clamped, never stored)
    nt_phase_bf3_ring<BM, BN, MSUB, NS, NS, NS - 1>(smem, ar, br, H, wm * 32 * MSUB, b_row0, acc);
  else if constexpr (BF3)
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

}  // namespace cmhse
