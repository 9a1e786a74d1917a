// nt_core.hpp — exact-fp32 MFMA "NT" tile main loop for gfx950 (CDNA4).
//
// Every contraction on the CMHSE hot path has the same shape: C[m][n] = sum_k A[m][k] * B[n][k]
// with BOTH operands K-contiguous rows (hidden states x weight rows, embeddings x embeddings).
// This header implements that tile loop once; the GRU step, the attention-energy GEMM, the
// similarity/rank kernel and the cosine-sim kernel differ only in how rows are addressed
// (loaders) and in their epilogues.
//
// Machine mapping (MI355X_MICROARCH.md / cdna_hip_programming.md §3):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise a k-ordered fmaf chain), 64 cycles per issue per
//     SIMD, one A and one B VGPR per lane: lane l supplies A[i = l&31][k = l>>5] and
//     B[k = l>>5][j = l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//   * the k index inside an MFMA is a free permutation as long as A and B agree, so each lane
//     reads a float4 (ds_read_b128) = 4 consecutive k for its lane-half and feeds 4 MFMAs:
//     within a block of 8 k, half h = l>>5 owns k = 8*kb + 4*h + {0,1,2,3}.
//   * LDS tiles are [row][BK=16] with a row stride of 20 floats (80 B): 16-byte aligned for
//     b128 access and conflict-free for the 16-lane b128 read groups.
//   * 256-thread workgroups = 4 waves arranged 2 (M) x 2 (N); global -> register prefetch of
//     chunk c+1 is issued before the MFMAs of chunk c, LDS is double-buffered, one barrier/chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmhse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;       // floats of K per LDS chunk
constexpr int kLdsLd = 20;    // LDS row stride in floats
constexpr int kThreads = 256; // 4 waves

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Guarded 4-float load of row `p` at k..k+3 (< klim).  `vec` = row base 16-byte aligned & K%4==0.
__device__ __forceinline__ float4 load_row4(const float* __restrict__ p, int k, int klim,
                                            bool vec) {
  if (p == nullptr || k >= klim) return zero4();
  if (vec) return *reinterpret_cast<const float4*>(p + k);
  float4 v;
  v.x = p[k];
  v.y = (k + 1 < klim) ? p[k + 1] : 0.f;
  v.z = (k + 2 < klim) ? p[k + 2] : 0.f;
  v.w = (k + 3 < klim) ? p[k + 3] : 0.f;
  return v;
}

template <int BM, int BNR>
struct TileSmem {
  static constexpr int kAFloats = BM * kLdsLd;
  static constexpr int kBFloats = BNR * kLdsLd;
  static constexpr int kFloats = 2 * (kAFloats + kBFloats);
  static constexpr size_t kBytes = sizeof(float) * kFloats;
  __device__ static float* a(float* base, int buf) { return base + buf * kAFloats; }
  __device__ static float* b(float* base, int buf) {
    return base + 2 * kAFloats + buf * kBFloats;
  }
};

// One phase of the K loop.
//   BM, BNR      rows of the A / B tile staged per chunk
//   MSUB, NSUB   32-row sub-tiles per wave in M / N
//   NACC         accumulators per M sub-tile the caller owns; sub-tile ns of this phase adds into
//                accumulator (ns == NSUB-1 ? LAST : ns) — lets the GRU share r/z accumulators
//                between its x phase and its h phase while keeping the two n-gate terms apart.
//   arow[i]/brow[i]  per-thread row base pointers (nullptr = zero row) for the rows this thread
//                stages: row = (tid >> 2) + 64*i, 16-byte slot = tid & 3.
template <int BM, int BNR, int MSUB, int NSUB, int NACC, int LAST>
__device__ __forceinline__ void nt_phase(float* smem, const float* const (&arow)[BM / 64],
                                         const float* const (&brow)[BNR / 64], int K, bool avec,
                                         bool bvec, int a_row0, const int (&b_row0)[NSUB],
                                         f32x16 (&acc)[MSUB][NACC]) {
  using SM = TileSmem<BM, BNR>;
  constexpr int AP = BM / 64, BP = BNR / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int srow = tid >> 2;
  const int sk = (tid & 3) * 4;
  const int nchunks = (K + kBK - 1) / kBK;
  if (nchunks == 0) return;

  float4 ra[AP], rb[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) ra[i] = load_row4(arow[i], sk, K, avec);
#pragma unroll
  for (int i = 0; i < BP; ++i) rb[i] = load_row4(brow[i], sk, K, bvec);
  __syncthreads();  // previous phase / kernel section finished reading LDS
#pragma unroll
  for (int i = 0; i < AP; ++i)
    *reinterpret_cast<float4*>(SM::a(smem, 0) + (srow + 64 * i) * kLdsLd + sk) = ra[i];
#pragma unroll
  for (int i = 0; i < BP; ++i)
    *reinterpret_cast<float4*>(SM::b(smem, 0) + (srow + 64 * i) * kLdsLd + sk) = rb[i];
  __syncthreads();

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    const bool more = (c + 1 < nchunks);
    if (more) {
      const int k = (c + 1) * kBK + sk;
#pragma unroll
      for (int i = 0; i < AP; ++i) ra[i] = load_row4(arow[i], k, K, avec);
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = load_row4(brow[i], k, K, bvec);
    }
    const float* As = SM::a(smem, cur);
    const float* Bs = SM::b(smem, cur);
#pragma unroll
    for (int kb = 0; kb < kBK / 8; ++kb) {
      float4 af[MSUB], bf[NSUB];
#pragma unroll
      for (int ms = 0; ms < MSUB; ++ms)
        af[ms] = *reinterpret_cast<const float4*>(As + (a_row0 + ms * 32 + frow) * kLdsLd +
                                                  kb * 8 + fk);
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns)
        bf[ns] = *reinterpret_cast<const float4*>(Bs + (b_row0[ns] + frow) * kLdsLd + kb * 8 +
                                                  fk);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms) {
          const float av = (j == 0) ? af[ms].x : (j == 1) ? af[ms].y : (j == 2) ? af[ms].z
                                                                                : af[ms].w;
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            const float bv = (j == 0) ? bf[ns].x : (j == 1) ? bf[ns].y : (j == 2) ? bf[ns].z
                                                                                  : bf[ns].w;
            constexpr int kLast = LAST;
            const int ai = (ns == NSUB - 1) ? kLast : ns;
            acc[ms][ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ms][ai], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      const int nxt = cur ^ 1;
#pragma unroll
      for (int i = 0; i < AP; ++i)
        *reinterpret_cast<float4*>(SM::a(smem, nxt) + (srow + 64 * i) * kLdsLd + sk) = ra[i];
#pragma unroll
      for (int i = 0; i < BP; ++i)
        *reinterpret_cast<float4*>(SM::b(smem, nxt) + (srow + 64 * i) * kLdsLd + sk) = rb[i];
    }
    __syncthreads();
  }
}

// Row / column owned by accumulator register r of lane `lane` inside a 32x32 sub-tile.
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

}  // namespace cmhse
