// tn_core.hpp — exact-fp32 MFMA "TN" tile loop:  C[m][n] = sum_k A[k][m] * B[k][n].
//
// The shape of every weight gradient on the path (dW = dGates^T . X summed over all packed
// (sequence, step) rows k) and of the loss gradients (G^T . im, G . s): the reduction index is the
// ROW index of both operands.  Tiles are staged [k][m] in LDS exactly as they lie in memory
// (coalesced float4 along m / n), and the MFMA fragments (A[i = lane&31][k = lane>>5]) are read
// with ds_read_b32 — 32 consecutive lanes read 32 consecutive floats, conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nt_core.hpp"

namespace cmhse {

constexpr int kTnBK = 16;

template <int BM, int BN>
struct TnSmem {
  static constexpr int kLdA = BM + 4, kLdB = BN + 4;
  static constexpr int kAFloats = kTnBK * kLdA, kBFloats = kTnBK * kLdB;
  static constexpr size_t kBytes = sizeof(float) * 2 * (kAFloats + kBFloats);
};

// 4 floats of a k-row at columns c..c+3 (< ncols); zeros when the row is out of range.
template <bool VEC>
__device__ __forceinline__ float4 tn_load4(rowaddr_t row, bool row_ok, int c, int ncols) {
  float4 v = zero4();
  if (!row_ok) return v;
  if (VEC) {
    if (c < ncols) {
      const f32x4 g = *(gptr_f32x4)(row + static_cast<rowaddr_t>(c) * 4u);
      v = make_float4(g.x, g.y, g.z, g.w);
    }
  } else {
    gptr_f32 g = (gptr_f32)row;
    if (c < ncols) v.x = g[c];
    if (c + 1 < ncols) v.y = g[c + 1];
    if (c + 2 < ncols) v.z = g[c + 2];
    if (c + 3 < ncols) v.w = g[c + 3];
  }
  return v;
}

// A rows: a_base + k*lda (floats).  B rows: b_addr[k] when b_addr != nullptr else b_base + k*ldb.
// The workgroup computes C[m0 .. m0+BM) x [n0 .. n0+BN); 4 waves as 2 (M) x 2 (N).
template <int BM, int BN, bool VEC>
__device__ __forceinline__ void tn_mainloop(float* smem, const float* a_base, int64_t lda, int M,
                                            const float* b_base, int64_t ldb,
                                            const uint64_t* b_addr, int N, int64_t K, int m0,
                                            int n0, f32x16 (&acc)[BM / 64][BN / 64]) {
  using SM = TnSmem<BM, BN>;
  constexpr int MSUB = BM / 64, NSUB = BN / 64;
  constexpr int AF4 = BM / 4, BF4 = BN / 4;                  // float4 per k-row
  constexpr int AP = kTnBK * AF4 / kThreads, BP = kTnBK * BF4 / kThreads;
  static_assert(AP >= 1 && BP >= 1, "tile too small");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* As[2] = {smem, smem + SM::kAFloats};
  float* Bs[2] = {smem + 2 * SM::kAFloats, smem + 2 * SM::kAFloats + SM::kBFloats};
  const int64_t nchunks = (K + kTnBK - 1) / kTnBK;
  float4 ra[AP], rb[BP];

  auto load = [&](int64_t c) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx / AF4, col = (idx % AF4) * 4;
      const int64_t k = c * kTnBK + kr;
      ra[i] = tn_load4<VEC>(row_addr(a_base + k * lda + m0), k < K, col, M - m0);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx / BF4, col = (idx % BF4) * 4;
      const int64_t k = c * kTnBK + kr;
      const bool ok = k < K;
      rowaddr_t row = 0;
      if (ok) row = (b_addr != nullptr) ? b_addr[k] + static_cast<rowaddr_t>(n0) * 4u
                                        : row_addr(b_base + k * ldb + n0);
      rb[i] = tn_load4<VEC>(row, ok, col, N - n0);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int idx = tid + i * kThreads;
      *reinterpret_cast<float4*>(As[buf] + (idx / AF4) * SM::kLdA + (idx % AF4) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int idx = tid + i * kThreads;
      *reinterpret_cast<float4*>(Bs[buf] + (idx / BF4) * SM::kLdB + (idx % BF4) * 4) = rb[i];
    }
  };

  if (nchunks == 0) return;
  load(0);
  __syncthreads();
  store(0);
  __syncthreads();
  const int fi = lane & 31, fh = (lane >> 5) * 4;
  for (int64_t c = 0; c < nchunks; ++c) {
    const int cur = static_cast<int>(c & 1);
    if (c + 1 < nchunks) load(c + 1);
#pragma unroll
    for (int kb = 0; kb < kTnBK / 8; ++kb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb * 8 + fh + j;
        float av[MSUB], bv[NSUB];
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms) av[ms] = As[cur][k * SM::kLdA + wm * 32 * MSUB + ms * 32 + fi];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bv[ns] = Bs[cur][k * SM::kLdB + wn * 32 * NSUB + ns * 32 + fi];
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ms], bv[ns], acc[ms][ns], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) store(cur ^ 1);
    __syncthreads();
  }
}

}  // namespace cmhse
