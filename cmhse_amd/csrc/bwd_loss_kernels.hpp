// bwd_loss_kernels.hpp
//
// Backward of F.normalize, ContrastiveLoss (from the stored scores), GroupWiseContrastiveLoss's block expansion and
// decoder.loss.EuclideanLoss.  Included by bwd.hip only.
#pragma once

namespace cmhse {

// ---------------------------------------------------------------------------------------------
// F.normalize backward: dx = (g - y (y.g)) / max(||x||, eps), y = x / max(||x||, eps)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void l2norm_bwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ g,
                                                              float* __restrict__ dx, int cols) {
  const int64_t row = blockIdx.x;
  const float* xr = x + row * cols;
  const float* gr = g + row * cols;
  float ss = 0.f, sg = 0.f;
  for (int c = threadIdx.x; c < cols; c += kThreads) {
    ss += xr[c] * xr[c];
    sg += xr[c] * gr[c];
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
    s_dot = (p2[0] + p2[1] + p2[2] + p2[3]) * inv * inv;  // (y.g)/||x||
  }
  __syncthreads();
  const float inv = s_inv, d = s_dot;
  for (int c = threadIdx.x; c < cols; c += kThreads) dx[row * cols + c] = (gr[c] - xr[c] * d) * inv;
}

// ---------------------------------------------------------------------------------------------
// ContrastiveLoss backward: G = d loss / d scores (loss.py:94-117 differentiated), then
// d im = G . s and d s = G^T . im as TN GEMMs on G^T and G.
// ---------------------------------------------------------------------------------------------
struct LossBwdParams {
  const float* scores;  // [n, n]
  const float* gout;    // device scalar: upstream gradient
  int32_t n, max_violation, norm;
  float margin;
  int32_t* row_arg;  // [n] max_violation: argmax_j cost_s(i,j) (or -1 when the max is 0)
  int32_t* col_arg;  // [n] max_violation: argmax_i cost_im(i,j)
  float* row_cnt;    // [n] sum_j g_s(i,j)
  float* col_cnt;    // [n] sum_i g_im(i,j)
  float* G;          // [n, n]
  float* GT;         // [n, n]
  // several independent losses in one launch set (cmhse_contrastive_blocks_bwd): block
  // b = blockIdx.y has n = blk_off[b+1] - blk_off[b]; its scores / G / GT start b * blk_stride
  // floats into their buffers (leading dimension = its own n), its vectors b * vec_stride, its
  // upstream gradient is gout[b]
  const int32_t* blk_off;
  int64_t blk_stride;     // floats between the blocks of G / GT
  int64_t score_stride;   // floats between the blocks of the stored scores (the forward's layout)
  int32_t vec_stride;
};

// the parameters of block blockIdx.y (or the struct itself for a single loss)
__device__ __forceinline__ LossBwdParams loss_bwd_block(const LossBwdParams& q_) {
  LossBwdParams q = q_;
  if (q.blk_off != nullptr) {
    const int b = blockIdx.y;
    q.n = q.blk_off[b + 1] - q.blk_off[b];
    q.scores += b * q.score_stride;
    q.G += b * q.blk_stride;
    q.GT += b * q.blk_stride;
    q.row_arg += b * q.vec_stride;
    q.col_arg += b * q.vec_stride;
    q.row_cnt += b * q.vec_stride;
    q.col_cnt += b * q.vec_stride;
    q.gout += b;
  }
  return q;
}

// one wave per row (rows pass) or per column (columns pass): counts / argmax of violating entries
__global__ __launch_bounds__(kThreads) void loss_bwd_stats_kernel(const LossBwdParams q_) {
  const LossBwdParams q = loss_bwd_block(q_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = q.n;
  const int idx = blockIdx.x * 4 + wave;  // [0, 2n): rows then columns
  if (idx >= 2 * n) return;
  const bool is_row = idx < n;
  const int i = is_row ? idx : idx - n;
  const float dii = q.scores[static_cast<int64_t>(i) * n + i];
  float cnt = 0.f, best = 0.f;
  int arg = 0x7fffffff;
  for (int j = lane; j < n; j += 64) {
    if (j == i) continue;
    const float sv = is_row ? q.scores[static_cast<int64_t>(i) * n + j]
                            : q.scores[static_cast<int64_t>(j) * n + i];
    const float c = fmaxf(q.margin + sv - dii, 0.f);
    cnt += (c > 0.f) ? 1.f : 0.f;
    if (c > best) {  // first maximum along the reduced index
      best = c;
      arg = j;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  if (lane == 0) {
    const int a = (best > 0.f) ? arg : -1;
    const float c = q.max_violation ? ((best > 0.f) ? 1.f : 0.f) : cnt;
    if (is_row) {
      q.row_arg[i] = a;
      q.row_cnt[i] = c;
    } else {
      q.col_arg[i] = a;
      q.col_cnt[i] = c;
    }
  }
}

__global__ __launch_bounds__(kThreads) void loss_bwd_build_kernel(const LossBwdParams q_) {
  const LossBwdParams q = loss_bwd_block(q_);
  const int n = q.n;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= static_cast<int64_t>(n) * n) return;
  const int i = static_cast<int>(e / n), j = static_cast<int>(e % n);
  float scale = *q.gout;
  if (q.norm) scale /= static_cast<float>(static_cast<int64_t>(n) * n);
  float g;
  if (i == j) {
    g = -(q.row_cnt[i] + q.col_cnt[i]);
  } else if (q.max_violation) {
    g = ((q.row_arg[i] == j) ? 1.f : 0.f) + ((q.col_arg[j] == i) ? 1.f : 0.f);
  } else {
    const float sv = q.scores[e];
    const float cs = q.margin + sv - q.scores[static_cast<int64_t>(i) * n + i];
    const float ci = q.margin + sv - q.scores[static_cast<int64_t>(j) * n + j];
    g = ((cs > 0.f) ? 1.f : 0.f) + ((ci > 0.f) ? 1.f : 0.f);
  }
  g *= scale;
  q.G[e] = g;
  q.GT[static_cast<int64_t>(j) * n + i] = g;
}

// GroupWiseContrastiveLoss backward: spread G_red[i][j] over block (i, j) of d scores (uniformly
// for the block mean, onto the arg-max for the block max); writes dS and dS^T.
struct ExpandParams {
  const float* g_red;  // [B, B]
  const int32_t* arg;  // [B, B]
  const int32_t* row_off;
  const int32_t* col_off;
  float* dS;
  float* dST;
  int32_t n, B, use_max;
};

__global__ __launch_bounds__(kThreads) void groupwise_expand_kernel(const ExpandParams q) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int r0 = q.row_off[bi], r1 = q.row_off[bi + 1], c0 = q.col_off[bj], c1 = q.col_off[bj + 1];
  const int w = c1 - c0, cnt = (r1 - r0) * w;
  const float g = q.g_red[bi * q.B + bj];
  const int a = q.arg[bi * q.B + bj];
  for (int e = threadIdx.x; e < cnt; e += kThreads) {
    const int r = r0 + e / w, c = c0 + e % w;
    const float v = q.use_max ? ((r * q.n + c == a) ? g : 0.f) : g / static_cast<float>(cnt);
    q.dS[static_cast<int64_t>(r) * q.n + c] = v;
    q.dST[static_cast<int64_t>(c) * q.n + r] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// EuclideanLoss (decoder/loss.py:17-26): per-row distances, fixed-order fp64 total
// ---------------------------------------------------------------------------------------------
struct EuclidParams {
  const float* a;
  const float* b;
  const uint64_t* b_rows;
  float* dist;  // [rows]
  float* d_a;   // backward
  const float* gout;
  float* loss;
  int32_t rows, cols, norm;
};

__global__ __launch_bounds__(kThreads) void euclid_rows_kernel(const EuclidParams q, int backward) {
  const int64_t r = blockIdx.x;
  const float* ar = q.a + r * q.cols;
  const float* br = q.b_rows ? reinterpret_cast<const float*>(q.b_rows[r]) : q.b + r * q.cols;
  float ss = 0.f;
  for (int c = threadIdx.x; c < q.cols; c += kThreads) {
    const float d = ar[c] - br[c];
    ss += d * d;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
  __shared__ float part[4];
  __shared__ float s_d;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) s_d = sqrtf(part[0] + part[1] + part[2] + part[3]);
  __syncthreads();
  const float dist = s_d;
  if (!backward) {
    if (threadIdx.x == 0) q.dist[r] = dist;
    return;
  }
  // d sqrt(ss) / d a = (a - b) / dist: no epsilon, like autograd through torch.sqrt upstream
  // (decoder/loss.py:21) — an exactly reconstructed row (dist == 0) gives NaN there and here
  float sc = *q.gout / dist;
  if (q.norm) sc /= static_cast<float>(q.rows);
  for (int c = threadIdx.x; c < q.cols; c += kThreads) q.d_a[r * q.cols + c] = (ar[c] - br[c]) * sc;
}

// Sum of the row distances in a FIXED order (bitwise reproducible): thread i adds rows i, i + 256,
// ... in double, then a pairwise LDS tree.  (--lowest_reconstruct_loss sums one row per frame /
// word of the batch, 1e4 and more: a single-thread dependent load chain took milliseconds.)
__global__ __launch_bounds__(kThreads) void euclid_final_kernel(const EuclidParams q) {
  __shared__ double part[kThreads];
  double t = 0.0;
  for (int r = threadIdx.x; r < q.rows; r += kThreads) t += q.dist[r];
  part[threadIdx.x] = t;
  __syncthreads();
  for (int w = kThreads / 2; w >= 1; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double total = part[0];
    if (q.norm) total /= q.rows;
    *q.loss = static_cast<float>(total);
  }
}

}  // namespace cmhse
