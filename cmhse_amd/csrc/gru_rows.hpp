// gru_rows.hpp
//
// Row kernels around the encoders: F.normalize, the embedding gather, host <-> HBM hand-over of
// feature / embedding rows (pull_steps, push_bytes), the bf16x3 operand pre-split, collate_fn's
// padding.  Included by gru.hip only.
#pragma once

namespace cmhse {

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

// Both copy kernels below run BESIDE the step chain, whose 128-row workgroups hold 2 x 232 of a SIMD's
// 512 vector registers: a copy wave that needs at most 48 fits into what is left and costs the chain
// no workgroup slot; one that needs more (round 5's pull: 68 + 144 B of scratch for its `float4 v[8]`,
// round 6's first push: 56) displaces a chain workgroup wherever it lands (the level-2 chain went
// from 3.1 to 5.1 ms beside 32 such waves, profiles/r06_api_path.txt).  Hence: single-wave workgroups,
// wave-uniform values forced into scalar registers, eight named 16-byte registers in flight per lane
// with compile-time strides (one address register pair per direction) — 48 and 44 registers, no scratch.
__device__ __forceinline__ uint64_t wave_uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// n16 16-byte units from src to dst (both 16-byte aligned, wave-uniform) by ONE wavefront.
__device__ __forceinline__ void wave_copy16(uint64_t src, uint64_t dst, uint32_t n16) {
  constexpr uint32_t chunk = 8 * 64;
  const uint32_t nchunks = n16 / chunk;
  const float4* sp = reinterpret_cast<const float4*>(src) + threadIdx.x;
  float4* dp = reinterpret_cast<float4*>(dst) + threadIdx.x;
  for (uint32_t c = 0; c < nchunks; ++c, sp += chunk, dp += chunk) {
    const float4 v0 = sp[0], v1 = sp[64], v2 = sp[128], v3 = sp[192];
    const float4 v4 = sp[256], v5 = sp[320], v6 = sp[384], v7 = sp[448];
    dp[0] = v0; dp[64] = v1; dp[128] = v2; dp[192] = v3;
    dp[256] = v4; dp[320] = v5; dp[384] = v6; dp[448] = v7;
  }
  for (uint32_t i = nchunks * chunk + threadIdx.x; i < n16; i += 64)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}

__global__ __launch_bounds__(64) void pull_steps_kernel(const PullParams p) {
  for (int s = blockIdx.x; s < p.n_active; s += gridDim.x) {
    const int len = __builtin_amdgcn_readfirstlane(p.lens[s]);
    const int te = (p.t1 < len) ? p.t1 : len;
    if (te <= p.t0) break;     // sorted longest first: every later sequence is shorter still
    const uint64_t off = static_cast<uint64_t>(p.t0) * p.row_floats * 4u;
    const uint64_t n = static_cast<uint64_t>(te - p.t0) * p.row_floats;
    const uint64_t src = wave_uniform64(p.src_rows[s]) + off, dst = wave_uniform64(p.dst_rows[s]) + off;
    if (((src | dst) & 15u) == 0 && (n & 3u) == 0) {
      wave_copy16(src, dst, static_cast<uint32_t>(n >> 2));   // (one sequence's steps [t0, t1): far below 2^32 units)
    } else {
      const float* sp = reinterpret_cast<const float*>(src);
      float* dp = reinterpret_cast<float*>(dst);
      for (uint64_t i = threadIdx.x; i < n; i += 64) dp[i] = sp[i];
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
__global__ __launch_bounds__(64) void push_bytes_kernel(const float4* __restrict__ src,
                                                       float4* __restrict__ dst, size_t n16,
                                                       const unsigned char* __restrict__ src_tail,
                                                       unsigned char* __restrict__ dst_tail, int tail) {
  // wavefront w of G owns the chunks w, w + G, ... of 8 x 64 units; the first one also the ragged end
  constexpr size_t chunk = 8 * 64;
  const size_t nchunks = n16 / chunk;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float4* sp = src + c * chunk + threadIdx.x;
    float4* dp = dst + c * chunk + threadIdx.x;
    const float4 v0 = sp[0], v1 = sp[64], v2 = sp[128], v3 = sp[192];
    const float4 v4 = sp[256], v5 = sp[320], v6 = sp[384], v7 = sp[448];
    dp[0] = v0; dp[64] = v1; dp[128] = v2; dp[192] = v3;
    dp[256] = v4; dp[320] = v5; dp[384] = v6; dp[448] = v7;
  }
  if (blockIdx.x == 0) {
    for (size_t i = nchunks * chunk + threadIdx.x; i < n16; i += 64) dst[i] = src[i];
    if (static_cast<int>(threadIdx.x) < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
  }
}

// ---------------------------------------------------------------------------------------------
// Do two byte ranges hold the same bytes?  `a` may be page-locked HOST memory (read zero-copy over
// PCIe), `b` is device memory: evaluation.i2t / t2i use it to check that the NumPy arrays they are
// handed still hold what encode_data wrote before they report the ranking encode_data queued.  One
// pass at the link's rate, nothing staged on the device; *flag is OR-ed with 1 on the first difference.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rows_differ_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                        size_t n16, const unsigned char* __restrict__ a_tail,
                                                        const unsigned char* __restrict__ b_tail, int tail,
                                                        int32_t* __restrict__ flag) {
  constexpr size_t chunk = 4 * 64;
  const size_t nchunks = n16 / chunk;
  unsigned diff = 0;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const uint4* ap = a + c * chunk + threadIdx.x;
    const uint4* bp = b + c * chunk + threadIdx.x;
    const uint4 a0 = ap[0], a1 = ap[64], a2 = ap[128], a3 = ap[192];
    const uint4 b0 = bp[0], b1 = bp[64], b2 = bp[128], b3 = bp[192];
    diff |= (a0.x ^ b0.x) | (a0.y ^ b0.y) | (a0.z ^ b0.z) | (a0.w ^ b0.w);
    diff |= (a1.x ^ b1.x) | (a1.y ^ b1.y) | (a1.z ^ b1.z) | (a1.w ^ b1.w);
    diff |= (a2.x ^ b2.x) | (a2.y ^ b2.y) | (a2.z ^ b2.z) | (a2.w ^ b2.w);
    diff |= (a3.x ^ b3.x) | (a3.y ^ b3.y) | (a3.z ^ b3.z) | (a3.w ^ b3.w);
  }
  if (blockIdx.x == 0) {
    for (size_t i = nchunks * chunk + threadIdx.x; i < n16; i += 64) {
      const uint4 x = a[i], y = b[i];
      diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    if (static_cast<int>(threadIdx.x) < tail) diff |= static_cast<unsigned>(a_tail[threadIdx.x] ^ b_tail[threadIdx.x]);
  }
  if (__builtin_amdgcn_ballot_w64(diff != 0) != 0 && threadIdx.x == 0) atomicOr(flag, 1);
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

}  // namespace cmhse
