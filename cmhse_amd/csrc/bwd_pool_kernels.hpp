// bwd_pool_kernels.hpp
//
// Backward of the pooling and the small helpers around the BPTT: transposes, column sums, the packed-row address
// tables, max / last pooling scatter, attention-pool backward (the +1e-4 of layers.py:158-162 is in its
// denominator too), and the plain TN / NT products of the attention projection's gradients.  Included by bwd.hip only.
#pragma once

namespace cmhse {

// ---------------------------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void transpose_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, int R,
                                                             int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = in[static_cast<int64_t>(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) out[static_cast<int64_t>(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// Column sums over the packed rows: out[c] = sum_p w[p] * in[p][c] (w == NULL -> 1), two stages,
// both in a fixed order (bitwise reproducible): stage 1 sums kColsumRows-row slabs (64 columns per
// workgroup, one per lane, the 4 waves interleave the slab's rows), stage 2 adds the slabs.
constexpr int kColsumRows = 512;

__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ part,
                                                                  int64_t rows, int cols,
                                                                  int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * kColsumRows;
  const int64_t r1 = (r0 + kColsumRows < rows) ? r0 + kColsumRows : rows;
  float s = 0.f;
  if (c < cols)
    for (int64_t p = r0 + wave; p < r1; p += 4) s += (w ? w[p] : 1.0f) * in[p * ld + c];
  __shared__ float red[4][64];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols)
    part[static_cast<int64_t>(blockIdx.y) * cols + c] =
        red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

__global__ __launch_bounds__(kThreads) void colsum_final_kernel(const float* __restrict__ part,
                                                                float* __restrict__ out,
                                                                int slabs, int cols) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int i = 0; i < slabs; ++i) s += part[static_cast<int64_t>(i) * cols + c];
  out[c] = s;
}

// Per packed row p = (t, s): addresses of x_{t,s} and of h_{t-1,s} (a zero row when there is none).
struct RowAddrParams {
  const uint64_t* x_rows;
  const uint64_t* tok_rows;
  const float* emb;
  const uint64_t* h0_rows;
  const int32_t* step_off;
  const float* hs;
  const float* zero_row;
  uint64_t* xaddr;
  uint64_t* hpaddr;
  uint64_t* hsaddr;   // address of h_{t,s} itself (the B rows of dW_lin), or NULL
  int32_t* p_t;
  int32_t Tmax, I, H, vocab, x_step;
  int64_t sum_T;
};

__global__ void row_addr_kernel(const RowAddrParams q) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= q.sum_T) return;
  int lo = 0, hi = q.Tmax;  // largest t with step_off[t] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (q.step_off[mid] <= p) lo = mid; else hi = mid;
  }
  const int t = lo, s = static_cast<int>(p - q.step_off[t]);
  if (q.p_t != nullptr) q.p_t[p] = t;
  if (q.tok_rows != nullptr) {
    long long tok = reinterpret_cast<const long long*>(q.tok_rows[s])[t];
    tok = tok < 0 ? 0 : (tok >= q.vocab ? q.vocab - 1 : tok);
    q.xaddr[p] = reinterpret_cast<uint64_t>(q.emb + tok * q.I);
  } else {
    q.xaddr[p] = q.x_rows[s] + static_cast<uint64_t>(t) * q.x_step * 4u;
  }
  if (q.hsaddr != nullptr) q.hsaddr[p] = reinterpret_cast<uint64_t>(q.hs + p * q.H);
  if (q.hpaddr == nullptr) return;
  if (t > 0)
    q.hpaddr[p] = reinterpret_cast<uint64_t>(q.hs + (static_cast<int64_t>(q.step_off[t - 1]) + s) * q.H);
  else if (q.h0_rows != nullptr)
    q.hpaddr[p] = q.h0_rows[s];
  else
    q.hpaddr[p] = reinterpret_cast<uint64_t>(q.zero_row);
}

// ---------------------------------------------------------------------------------------------
// pooling backward: fills dpool[p][u] = d loss / d h_p[u] coming from the pooling
// ---------------------------------------------------------------------------------------------
struct PoolBwdParams {
  const float* dout;  // [S, H] indexed by out_row
  const int32_t* lens;
  const int32_t* out_row;
  const int32_t* step_off;
  const int32_t* argmax;
  float* dpool;
  int32_t S, H, mode;
};

__global__ __launch_bounds__(kThreads) void pool_scatter_bwd_kernel(const PoolBwdParams q) {
  const int s = blockIdx.x;
  if (q.mode == CMHSE_POOL_ALL) {  // every hidden state is an output row: out_row[s] + t
    const int len = q.lens[s];
    for (int t = 0; t < len; ++t) {
      const float* g = q.dout + (static_cast<int64_t>(q.out_row[s]) + t) * q.H;
      float* d = q.dpool + (static_cast<int64_t>(q.step_off[t]) + s) * q.H;
      for (int u = threadIdx.x; u < q.H; u += kThreads) d[u] = g[u];
    }
    return;
  }
  const float* g = q.dout + static_cast<int64_t>(q.out_row[s]) * q.H;
  for (int u = threadIdx.x; u < q.H; u += kThreads) {
    const int t = (q.mode == CMHSE_POOL_MAX) ? q.argmax[static_cast<int64_t>(s) * q.H + u]
                                             : (q.lens[s] - 1);
    q.dpool[(static_cast<int64_t>(q.step_off[t]) + s) * q.H + u] = g[u];
  }
}

// attention: a_t = exp(e_t)/(sum exp + 1e-4); da_t = g . h_t; de_t = a_t (da_t - sum a da);
// dpool[p] = a_t g.  One workgroup per sequence.
struct AttnBwdParams {
  const float* dout;
  const float* hs;
  const float* e_part;
  const int32_t* lens;
  const int32_t* out_row;
  const int32_t* step_off;
  float* dpool;
  float* de;  // [sumT]
  int64_t rows;
  int32_t H, n_tiles;
};

// One workgroup per sequence; its four waves take the time steps round-robin and need no barrier
// per step (a wave reduces its own dot products with shuffles, and pass 2 revisits exactly the
// steps the same wave handled in pass 1, so it reads back its own da_t).  The previous form walked
// the steps one by one with two workgroup barriers each: 0.22-0.34 ms per call at T <= 80, at the
// head of each tower's backward pass.
constexpr int kPoolBwdThreads = 1024;   // 16 waves: a sequence of 80 steps is 5 steps per wave
__global__ __launch_bounds__(kPoolBwdThreads) void attn_pool_bwd_kernel(const AttnBwdParams q) {
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = q.lens[s], H = q.H;
  const float* g = q.dout + static_cast<int64_t>(q.out_row[s]) * H;
  constexpr int NT = kPoolBwdThreads, NW = NT / 64;
  __shared__ float red[NT];
  __shared__ float s_den, s_c;
  __shared__ float wpart[NW];
  auto energy = [&](int t) {
    const int64_t row = static_cast<int64_t>(q.step_off[t]) + s;
    float e = 0.f;
    for (int k = 0; k < q.n_tiles; ++k) e += q.e_part[k * q.rows + row];
    return e;
  };
  float part = 0.f;
  for (int t = tid; t < len; t += NT) part += expf(energy(t));
  red[tid] = part;
  __syncthreads();
  if (tid == 0) {
    float d = 0.f;
    const int used = len < NT ? len : NT;      // threads past `len` hold 0
    for (int i = 0; i < used; ++i) d += red[i];
    s_den = d + 0.0001f;
  }
  __syncthreads();
  const float den = s_den;
  // pass 1: da_t = g . h_t (one wave per step) -> de scratch holds da_t; c = sum_t a_t da_t
  float c_acc = 0.f;
  for (int t = wave; t < len; t += NW) {
    const int64_t row = static_cast<int64_t>(q.step_off[t]) + s;
    const float* hr = q.hs + row * H;
    float d = 0.f;
    for (int u = lane; u < H; u += 64) d += g[u] * hr[u];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o, 64);
    const float a = expf(energy(t)) / den;
    if (lane == 0) q.de[row] = d;  // temporarily da_t
    c_acc += a * d;
  }
  if (lane == 0) wpart[wave] = c_acc;
  __syncthreads();
  if (tid == 0) {
    float cs = 0.f;
    for (int w = 0; w < NW; ++w) cs += wpart[w];
    s_c = cs;
  }
  __syncthreads();
  const float c = s_c;
  // pass 2: de_t and dpool rows (same wave -> same steps as in pass 1)
  for (int t = wave; t < len; t += NW) {
    const int64_t row = static_cast<int64_t>(q.step_off[t]) + s;
    const float a = expf(energy(t)) / den;
    float* dp = q.dpool + row * H;
    for (int u = lane; u < H; u += 64) dp[u] = a * g[u];
    if (lane == 0) q.de[row] = a * (q.de[row] - c);
  }
}

// du[p][n] = de[p] * w_att[n] * (1 - v[p][n]^2)
__global__ __launch_bounds__(kThreads) void attn_du_kernel(const float* __restrict__ de,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ w_att,
                                                           float* __restrict__ du, int64_t rows,
                                                           int H) {
  const int64_t p = blockIdx.x;
  const float d = de[p];
  for (int n = threadIdx.x; n < H; n += kThreads) {
    const float tv = v[p * H + n];
    du[p * H + n] = d * w_att[n] * (1.0f - tv * tv);
  }
}

// ---------------------------------------------------------------------------------------------
// generic GEMM kernels
// ---------------------------------------------------------------------------------------------
struct TnParams {
  const float* a;  // [K, lda], columns m
  int64_t lda;
  const float* b;  // [K, ldb] or rows through b_addr
  int64_t ldb;
  const uint64_t* b_addr;
  float* c;  // [M, ldc]
  int64_t ldc;
  int32_t M, N, n_tiles;
  int64_t K;
  const float* scale;  // optional device scalar multiplied into C
  // several products in one launch (blockIdx.y = block b, rows blk_off[b] .. blk_off[b+1] of the
  // caller's row-blocked operands): A is the block's [n_b, n_b] matrix at a + b * blk_stride
  // (lda = n_b), B / C are rows blk_off[b].. of b / c; M = K = n_b
  const int32_t* blk_off;
  int64_t blk_stride;
};

template <bool VEC>
__global__ __launch_bounds__(kThreads) void gemm_tn_kernel(const TnParams q_) {
  TnParams q = q_;
  if (q.blk_off != nullptr) {
    const int b = blockIdx.y, off = q.blk_off[b], nb = q.blk_off[b + 1] - off;
    q.a += b * q.blk_stride;
    q.lda = nb;
    q.b += static_cast<int64_t>(off) * q.ldb;
    q.c += static_cast<int64_t>(off) * q.ldc;
    q.M = nb;
    q.K = nb;
    if ((blockIdx.x / q.n_tiles) * 128 >= nb) return;   // M tile outside this (smaller) block
  }
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = (blockIdx.x % q.n_tiles) * BN, m0 = (blockIdx.x / q.n_tiles) * BM;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  tn_mainloop<BM, BN, VEC>(smem, q.a, q.lda, q.M, q.b, q.ldb, q.b_addr, q.N, q.K, m0, n0, acc);
  const float sc = q.scale ? *q.scale : 1.0f;
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + ms * 32 + acc_row(r, lane);
        const int n = n0 + wn * 64 + ns * 32 + acc_col(lane);
        if (m < q.M && n < q.N) q.c[static_cast<int64_t>(m) * q.ldc + n] = sc * acc[ms][ns][r];
      }
}

// C[m][n] = sum_k A[m][k] B[n][k]; output row m goes to c_addr[m] (or c + m*ldc);
// mode 0 store, 1 accumulate (+=), 2 atomic add (rows may repeat: embedding-table scatter),
// 3 store the partial product of K segment blockIdx.y at c + blockIdx.y * M * ldc (dense scratch;
// splitk_reduce_kernel adds the segments in a fixed order).
struct NtOutParams {
  const float* a;  // [M, lda]
  int64_t lda;
  const float* b;  // [N, ldb]
  int64_t ldb;
  float* c;
  int64_t ldc;
  const uint64_t* c_addr;
  int32_t M, N, K, n_tiles, mode;
  int32_t k_seg;   // split-K: blockIdx.y takes k in [y*k_seg, min(K, (y+1)*k_seg)); 0 = no split
};

template <bool VEC>
__global__ __launch_bounds__(kThreads) void gemm_nt_out_kernel(const NtOutParams q) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, srow = tid >> 2;
  const int n0 = (blockIdx.x % q.n_tiles) * BN, m0 = (blockIdx.x / q.n_tiles) * BM;
  rowaddr_t ar[2], br[2];
  bool av[2], bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + srow + 64 * i, n = n0 + srow + 64 * i;
    av[i] = m < q.M;
    bv[i] = n < q.N;
    ar[i] = row_addr(q.a + static_cast<int64_t>(av[i] ? m : 0) * q.lda);
    br[i] = row_addr(q.b + static_cast<int64_t>(bv[i] ? n : 0) * q.ldb);
  }
  int K = q.K;
  if (q.k_seg > 0) {
    const int k0 = static_cast<int>(blockIdx.y) * q.k_seg;
    K = (q.K - k0 < q.k_seg) ? (q.K - k0) : q.k_seg;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ar[i] += static_cast<rowaddr_t>(k0) * 4u;
      br[i] += static_cast<rowaddr_t>(k0) * 4u;
    }
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const int b_row0[2] = {wn * 64, wn * 64 + 32};
  nt_phase<BM, BN, 2, 2, 2, 1, VEC>(smem, ar, av, br, bv, K, wm * 64, b_row0, acc);
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + ms * 32 + acc_row(r, lane);
      if (m >= q.M) continue;
      float* crow = (q.c_addr && q.mode != 3) ? reinterpret_cast<float*>(q.c_addr[m])
                                              : q.c + static_cast<int64_t>(m) * q.ldc;
      if (q.mode == 3) crow += static_cast<int64_t>(blockIdx.y) * q.M * q.ldc;
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        const int n = n0 + b_row0[ns] + acc_col(lane);
        if (n >= q.N) continue;
        const float v = acc[ms][ns][r];
        if (q.mode == 0 || q.mode == 3) crow[n] = v;
        else if (q.mode == 1) crow[n] += v;
        else atomicAdd(crow + n, v);
      }
    }
}

// out row m (at c_addr[m], or c + m * ldc) = [its old value +] part[0][m] + part[1][m] + ... in segment
// order (bitwise reproducible)
__global__ __launch_bounds__(kThreads) void splitk_reduce_kernel(const float* __restrict__ part,
                                                                 const uint64_t* __restrict__ c_addr,
                                                                 float* __restrict__ c, int64_t ldc, int add,
                                                                 int splits, int M, int N) {
  const int m = blockIdx.x;
  float* dst = c_addr ? reinterpret_cast<float*>(c_addr[m]) : c + m * ldc;
  for (int n = threadIdx.x; n < N; n += kThreads) {
    float s = 0.f;
    for (int y = 0; y < splits; ++y) s += part[(static_cast<int64_t>(y) * M + m) * N + n];
    dst[n] = add ? dst[n] + s : s;
  }
}

}  // namespace cmhse
