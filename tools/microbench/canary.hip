// (1) canary_kernel — bystander canary: small workgroups (256 threads, 3 KB of LDS, ~40 live VGPRs) that fill their LDS and
// registers with patterns and re-check them for a while, re-read a global pattern buffer through the
// vector L1, and recompute a little floating-point math (expf, division) that must repeat bit for bit.
// Run on one stream while a suspect kernel runs on another: any mismatch means a CO-RESIDENT
// workgroup's state, loads or arithmetic were disturbed.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o canary.so canary.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void canary_kernel(uint32_t* report, const uint32_t* __restrict__ pattern,
                                                     uint32_t pattern_words, int iters) {
  __shared__ uint32_t l[768];
  const uint32_t tag = blockIdx.x * 1024u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 3; ++i) l[threadIdx.x + 256 * i] = tag ^ (0x9e3779b9u * (i + 1));
  uint32_t r[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) r[k] = tag * (k + 3) + 12345u;
  __syncthreads();
  uint32_t bad_l = 0, bad_r = 0, bad_g = 0, bad_f = 0, bad_b = 0, first = 0, firstv = 0;
  float ref = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const uint32_t v = *reinterpret_cast<volatile uint32_t*>(&l[threadIdx.x + 256 * i]);
      if (v != (tag ^ (0x9e3779b9u * (i + 1)))) ++bad_l;
    }
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      asm volatile("" : "+v"(r[k]));
      if (r[k] != tag * (k + 3) + 12345u) ++bad_r;
    }
    // global reads through L1: word w of the pattern buffer holds w * 2654435761 + 7
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w = ((tag * 37u + it * 1031u + q * 65537u) % (pattern_words / 4)) * 4;
      const uint4 v = *reinterpret_cast<const uint4*>(pattern + w);
      const uint32_t e0 = w * 2654435761u + 7u;
      if (v.x != e0 || v.y != (w + 1) * 2654435761u + 7u || v.z != (w + 2) * 2654435761u + 7u ||
          v.w != (w + 3) * 2654435761u + 7u) {
        if (!bad_g) { first = w; firstv = v.x; }
        ++bad_g;
      }
      acc += static_cast<float>(v.x & 1023u) * 1e-3f;
    }
    // arithmetic that must repeat bit for bit (same inputs every iteration)
    float f = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) f += expf(-0.37f * static_cast<float>((tag + q) & 31u)) / (1.0f + static_cast<float>(q));
    if (it == 0) ref = f;
    else if (__float_as_uint(f) != __float_as_uint(ref)) ++bad_f;
    if (acc < 0.f) report[63] = 1;      // keep acc alive
    // cross-thread hand-off through LDS behind a workgroup barrier (what a reduction kernel does)
    __shared__ uint32_t x[256];
    x[threadIdx.x] = tag + 977u * it;
    // ... and the attention-pooling kernel's shape: 64-bit values, read back as a BROADCAST (every lane the
    // same address) in a loop, used as an index
    __shared__ long long y[256];
    __shared__ float wgt[256];
    y[threadIdx.x] = static_cast<long long>(tag) * 4096 + it;
    wgt[threadIdx.x] = static_cast<float>((tag + it) & 255u);
    __syncthreads();
    {
      long long sum = 0;
      float fs = 0.f;
#pragma unroll 4
      for (int j = 0; j < 256; ++j) {
        sum += y[j] ^ j;
        fs += wgt[j];
      }
      long long want = 0;
      float fw = 0.f;
      for (int j = 0; j < 256; ++j) {
        want += (static_cast<long long>(blockIdx.x * 1024u + j) * 4096 + it) ^ j;
        fw += static_cast<float>((blockIdx.x * 1024u + j + it) & 255u);
      }
      if (sum != want || fs != fw) ++bad_b;
    }
    const uint32_t n1 = x[(threadIdx.x + 1) & 255], n2 = x[(threadIdx.x * 7 + 3) & 255];
    const uint32_t base_tag = blockIdx.x * 1024u;
    if (n1 != base_tag + ((threadIdx.x + 1) & 255) + 977u * it || n2 != base_tag + ((threadIdx.x * 7 + 3) & 255) + 977u * it) ++bad_b;
    __syncthreads();
    __builtin_amdgcn_s_sleep(4);
  }
  if (bad_l) atomicAdd(&report[0], 1u);
  if (bad_r) atomicAdd(&report[1], 1u);
  if (bad_g) {
    atomicAdd(&report[4], 1u);
    if (atomicAdd(&report[2], 1u) < 8) {
      const uint32_t slot = 8 + 4 * (atomicAdd(&report[3], 1u) & 7);
      report[slot] = blockIdx.x; report[slot + 1] = first; report[slot + 2] = firstv; report[slot + 3] = bad_g;
    }
  }
  if (bad_f) atomicAdd(&report[5], 1u);
  if (bad_b) atomicAdd(&report[6], 1u);
}

// (2) pkfma_canary_kernel (tools/pkfma_canary.py) — the one that found it, profiles/r05_bf16_mfma_bystander.txt.
// The attention-pooling kernel's inner loop on data whose sums are exact: acc[k] += w[j] * h[row_j][4 tid + k] with
// w = 1 and h small integers, rows and weights staged in LDS and read back as broadcasts, four 16-byte row loads in
// flight (the compiler emits v_pk_fma_f32 for the four accumulators, as in attn_pool_kernel).  Every workgroup
// re-derives the sums in integer arithmetic; a lost update shows up as a deficit that names the step it belongs to.
// report[0] = workgroup-iterations checked (low word), report[1] = mismatching (thread, component) pairs,
// report[8 + 8 i ..] = the first 14 mismatches: block, iteration, thread, component, got, want, step count, 0.
__global__ __launch_bounds__(256) void pkfma_canary_kernel(uint32_t* report, const float* __restrict__ hs, int rows,
                                                           int len, int iters) {
  __shared__ float s_w[256];
  __shared__ long long s_row[256];
  const int tid = threadIdx.x, u = 4 * tid;
  uint32_t bad = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (tid < len) {
      s_row[tid] = static_cast<long long>((blockIdx.x * 131u + it * 17u + tid * 29u) % static_cast<uint32_t>(rows));
      s_w[tid] = 1.0f;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int j = 0; j < len; ++j) {
      const float4 h = *reinterpret_cast<const float4*>(hs + s_row[j] * 1024 + u);
      const float wgt = s_w[j];
      a0 += wgt * h.x;
      a1 += wgt * h.y;
      a2 += wgt * h.z;
      a3 += wgt * h.w;
    }
    uint32_t e[4] = {0, 0, 0, 0};
    for (int j = 0; j < len; ++j) {
      const uint32_t r = static_cast<uint32_t>(s_row[j]);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] += (r * 5u + (u + k) * 3u) & 31u;
    }
    const float got[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (got[k] != static_cast<float>(e[k])) {
        ++bad;
        const uint32_t slot = atomicAdd(&report[2], 1u);
        if (slot < 14) {
          uint32_t* q = report + 8 + 8 * slot;
          q[0] = blockIdx.x; q[1] = it; q[2] = tid; q[3] = k; q[4] = __float_as_uint(got[k]); q[5] = e[k]; q[6] = len;
        }
      }
  }
  if (tid == 0) atomicAdd(&report[0], static_cast<uint32_t>(iters));
  if (bad) atomicAdd(&report[1], bad);
}

__global__ void fill_rows_pattern(float* p, uint32_t n) {      // hs[row][c] = (5 row + 3 c) & 31, 1024 columns
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = static_cast<float>(((i >> 10) * 5u + (i & 1023u) * 3u) & 31u);
}

extern "C" int pkfma_canary_fill(float* hs, uint32_t rows, void* stream) {
  const uint32_t n = rows * 1024u;
  hipLaunchKernelGGL(fill_rows_pattern, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), hs, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int pkfma_canary_launch(uint32_t* report, const float* hs, int rows, int len, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(pkfma_canary_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), report, hs, rows, len, iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void fill_pattern(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i * 2654435761u + 7u;
}

extern "C" int canary_fill(uint32_t* pattern, uint32_t words, void* stream) {
  hipLaunchKernelGGL(fill_pattern, dim3((words + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pattern, words);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int canary_launch(uint32_t* report, const uint32_t* pattern, uint32_t words, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), report, pattern, words, iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
