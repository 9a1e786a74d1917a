// Stand-alone reproducer: lost VGPR updates in a bystander workgroup on gfx950 (profiles/r05_bf16_mfma_bystander.txt).
//
// A "victim" kernel repeats attn_pool_kernel's inner loop (acc[k] += w[j] * h[row_j][4 tid + k]: rows and weights
// broadcast from LDS, four 16-byte row loads in flight) on small integers, so every sum is exact and is re-derived in
// integer arithmetic; its four accumulators are updated by v_pk_fma_f32 (what the compiler emits) or, in the other
// variants, by v_fmac_f32 / v_pk_mul+v_pk_add / v_pk_add / v_fma_f64.  An "offender" kernel runs on a second stream at
// the same time.  Sections, in the order they print:
//   1. the product's tile loops as offender (nt_core.hpp: nt_phase_bf3 with and without the in-register split, nt_phase);
//      compile with -DCMHSE_BF3_MFMA_32X32X16 to put the single gfx950 instruction back into nt_phase_bf3
//   2. synthetic mixes: four MFMAs + N VALU instructions of one kind per loop trip, LDS-fed operands, barriers
//   3. which victim instruction loses updates (beside the tile loop and beside the worst synthetic mix)
//   4. which matrix instruction does it (four MFMAs of one flavour + 24 / 48 v_fmac_f32 per trip)
//   5. single instruction kinds in a loop x the offender's VGPR allocation (all clean)
// Result on MI355X / ROCm 7.2: only v_pk_fma_f32 is hit, only lanes 48-63, only beside the matrix instructions gfx950
// added (32x32x16 bf16/f16, 16x16x32 bf16, 32x32x32 i8) when they are interleaved with other work.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pkfma_lost_update pkfma_lost_update.hip
//   ./pkfma_lost_update [reps=4] [grid of the synthetic offenders=256] [grid of the tile offender=64] [a 4th argument: skip section 5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../cmhse_amd/csrc/nt_core.hpp"   // the product's tile loops (nt_phase, nt_phase_bf3)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- victim -------------------------------------------------------------------------------------------------
enum { VK_FMAC, VK_PK_FMA, VK_PK_MUL_ADD, VK_FMA_F64, VK_PK_ADD, VK_COUNT };
static const char* kVictimName[VK_COUNT] = {"v_fmac_f32", "v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32", "v_fma_f64", "v_pk_add_f32 (h only)"};
template <int PACKED>
__global__ __launch_bounds__(256) void victim_kernel(uint32_t* report, const float* __restrict__ hs, int rows, int len,
                                                     int iters) {
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
    double d0 = 0., d1 = 0., d2 = 0., d3 = 0.;
    f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll 4
    for (int j = 0; j < len; ++j) {
      const float4 h = *reinterpret_cast<const float4*>(hs + s_row[j] * 1024 + u);
      const float wgt = s_w[j];
      if (PACKED == VK_PK_FMA) {
        a0 += wgt * h.x;
        a1 += wgt * h.y;
        a2 += wgt * h.z;
        a3 += wgt * h.w;
      } else if (PACKED == VK_PK_MUL_ADD) {
        f32x2 h01 = {h.x, h.y}, h23 = {h.z, h.w}, w2 = {wgt, wgt}, t01, t23;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t01) : "v"(h01), "v"(w2));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t23) : "v"(h23), "v"(w2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p01) : "v"(t01));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p23) : "v"(t23));
      } else if (PACKED == VK_PK_ADD) {
        f32x2 h01 = {h.x, h.y}, h23 = {h.z, h.w};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p01) : "v"(h01));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p23) : "v"(h23));
      } else if (PACKED == VK_FMA_F64) {
        const double dw = wgt;
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(dw), "v"(static_cast<double>(h.x)));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d1) : "v"(dw), "v"(static_cast<double>(h.y)));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d2) : "v"(dw), "v"(static_cast<double>(h.z)));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d3) : "v"(dw), "v"(static_cast<double>(h.w)));
      } else {
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(wgt), "v"(h.x));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(wgt), "v"(h.y));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(wgt), "v"(h.z));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(wgt), "v"(h.w));
      }
    }
    uint32_t e[4] = {0, 0, 0, 0};
    for (int j = 0; j < len; ++j) {
      const uint32_t r = static_cast<uint32_t>(s_row[j]);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] += (r * 5u + (u + k) * 3u) & 31u;
    }
    if (PACKED == VK_PK_MUL_ADD || PACKED == VK_PK_ADD) { a0 = p01[0]; a1 = p01[1]; a2 = p23[0]; a3 = p23[1]; }
    if (PACKED == VK_FMA_F64) { a0 = static_cast<float>(d0); a1 = static_cast<float>(d1); a2 = static_cast<float>(d2); a3 = static_cast<float>(d3); }
    const float got[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (got[k] != static_cast<float>(e[k])) {
        ++bad;
        const uint32_t slot = atomicAdd(&report[2], 1u);
        if (slot < 6) {
          uint32_t* q = report + 8 + 8 * slot;
          q[0] = blockIdx.x; q[1] = it; q[2] = tid; q[3] = k; q[4] = __float_as_uint(got[k]); q[5] = e[k];
        }
        atomicOr(&report[3], 1u << (tid & 63) / 16);      // which 16-lane quarter of the wave
      }
  }
  if (tid == 0) atomicAdd(&report[0], static_cast<uint32_t>(iters));
  if (bad) atomicAdd(&report[1], bad);
}

__global__ void fill_rows_pattern(float* p, uint32_t n) {      // hs[row][c] = (5 row + 3 c) & 31, 1024 columns
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = static_cast<float>(((i >> 10) * 5u + (i & 1023u) * 3u) & 31u);
}

// ---- offenders ----------------------------------------------------------------------------------------------
enum { M_IDLE, M_MFMA_BF16, M_MFMA_F32, M_CVT_PK, M_PK_ADD, M_LDS, M_PK_FMA, M_MFMA_BF16_16, M_VALU_F32, M_GLOBAL, M_COUNT };
static const char* kModeName[M_COUNT] = {
    "idle (s_sleep)", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "v_cvt_pk_bf16_f32", "v_pk_add_f32",
    "ds_write_b128 / ds_read_b128", "v_pk_fma_f32", "v_mfma_f32_16x16x32_bf16", "v_fmac_f32", "global_load_dwordx4"};

// PAD: the workgroup's VGPR allocation is raised to PAD registers (a clobber of v[PAD-1]), so that what a co-resident
// victim wave gets is the part of the SIMD's 512-row register file BEHIND a large allocation, as beside a tile kernel.
template <int MODE, int PAD>
__global__ __launch_bounds__(256) void offender_kernel(float* sink, const float* __restrict__ src, int iters) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x;
  float r = 0.f;
  if (PAD == 200) asm volatile("v_mov_b32 v199, 0" ::: "v199");
  if (PAD == 224) asm volatile("v_mov_b32 v223, 0" ::: "v223");
  if (PAD == 232) asm volatile("v_mov_b32 v231, 0" ::: "v231");
  if (PAD == 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
  if (PAD == 248) asm volatile("v_mov_b32 v247, 0" ::: "v247");
  if (PAD == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (MODE == M_IDLE) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(8);
  } else if (MODE == M_MFMA_BF16 || MODE == M_MFMA_BF16_16) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = static_cast<short>(0x3f80 + ((tid + k) & 7)); b[k] = static_cast<short>(0x3f80 + ((tid * 3 + k) & 7)); }
    if (MODE == M_MFMA_BF16) {
      f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      r = c0[0] + c1[5] + c2[9] + c3[15];
    } else {
      f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else if (MODE == M_MFMA_F32) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float a = 1.0f + (tid & 3), b = 0.5f;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    r = c0[0] + c1[5] + c2[9] + c3[15];
  } else if (MODE == M_CVT_PK) {
    float x = 1.0f + tid * 1e-3f, y = 2.0f;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t pk;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(y));
        acc += pk;
        x += 1e-3f;
      }
    }
    r = static_cast<float>(acc & 0xffff);
  } else if (MODE == M_PK_ADD) {
    f32x2 x = {1.0f + tid, 2.0f}, y = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    r = x[0] + x[1];
  } else if (MODE == M_PK_FMA) {
    f32x2 x = {1.0f + tid, 2.0f}, y = {1e-3f, 2e-3f}, z = {0.999f, 1.001f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(z), "v"(y));
    }
    r = x[0] + x[1];
  } else if (MODE == M_VALU_F32) {
    float x = 1.0f + tid, y = 1e-3f, z = 0.999f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(z), "v"(y));
    }
    r = x;
  } else if (MODE == M_LDS) {
    float4 v = make_float4(tid, 1.f, 2.f, 3.f);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lds[tid + 256 * k] = v;
        const float4 w = lds[(tid * 5 + 256 * k + 3) & 1023];
        v.x += w.y;
      }
    }
    r = v.x;
  } else if (MODE == M_GLOBAL) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < iters; ++i) {
      const float4 w = *reinterpret_cast<const float4*>(src + ((static_cast<size_t>(blockIdx.x) * 977u + i * 131u + tid) % (24576u * 256u)) * 4);
      v.x += w.x;
    }
    r = v.x;
  }
  if (r == 12345.678f) sink[tid] = r;
}

// The product's tile loop as the offender: a 128 x 256 tile over K = 1024, repeated; BF3 = nt_phase_bf3 (three
// v_mfma_f32_32x32x16_bf16 per fp32 product, A split in registers, operands through LDS), else nt_phase (fp32 MFMA).
template <int BF3>
__global__ __launch_bounds__(256) void tile_offender_kernel(float* sink, const float* __restrict__ a_rows,
                                                            const float* __restrict__ b_rows, int K, int iters) {
  using namespace cmhse;
  constexpr int MSUB = 2, BM = 64 * MSUB, BN = 256, NS = BN / 64;
  extern __shared__ __attribute__((aligned(16))) float tile_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, srow = tid >> 2;
  rowaddr_t ar[BM / 64], br[BN / 64];
  bool av[BM / 64], bv[BN / 64];
  for (int i = 0; i < BM / 64; ++i) { av[i] = true; ar[i] = row_addr(a_rows + static_cast<size_t>((blockIdx.x * BM + srow + 64 * i) % 24576) * 1024); }
  for (int i = 0; i < BN / 64; ++i) { bv[i] = true; br[i] = row_addr(b_rows + static_cast<size_t>(srow + 64 * i) * 1024); }
  f32x16 acc[MSUB][NS];
  for (int ms = 0; ms < MSUB; ++ms) for (int n = 0; n < NS; ++n) acc[ms][n] = zero16();
  int b_row0[NS];
  for (int ns = 0; ns < NS; ++ns) b_row0[ns] = wn * (BN / 2) + 32 * ns;
  for (int it = 0; it < iters; ++it) {
    if (BF3 == 1)
      nt_phase_bf3<BM, BN, MSUB, NS, NS, NS - 1, false>(tile_smem, ar, av, br, bv, K, wm * 32 * MSUB, b_row0, acc);
    else if (BF3 == 2)     // A pre-split: no conversion arithmetic in the loop
      nt_phase_bf3<BM, BN, MSUB, NS, NS, NS - 1, true>(tile_smem, ar, av, br, bv, K, wm * 32 * MSUB, b_row0, acc);
    else
      nt_phase<BM, BN, MSUB, NS, NS, NS - 1, true>(tile_smem, ar, av, br, bv, K, wm * 32 * MSUB, b_row0, acc);
  }
  float r = 0.f;
  for (int ms = 0; ms < MSUB; ++ms) for (int n = 0; n < NS; ++n) r += acc[ms][n][3];
  if (r == 12345.678f) sink[tid] = r;
}

// Synthetic mixes in ONE offender wave: four MFMAs (bf16 32x32x16 or fp32 32x32x2, constant or LDS-fed operands) and
// NV VALU instructions of one kind between them, per loop trip.
enum { V_NONE, V_PK_ADD, V_CVT_PK, V_FMAC, V_PK_FMA, V_PK_MUL, V_MOV, V_SNOP };
template <bool BF16, int VALU, int NV, bool LDS_FED, bool BARRIER>
__global__ __launch_bounds__(256) void combo_offender_kernel(float* sink, int iters) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = static_cast<short>(0x3f80 + ((tid + k) & 7)); b[k] = static_cast<short>(0x3f80 + ((tid * 3 + k) & 7)); }
  if (LDS_FED) { lds[tid] = __builtin_bit_cast(float4, a); lds[256 + tid] = __builtin_bit_cast(float4, b); __syncthreads(); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  f32x2 x = {1.0f + tid, 2.0f}, y = {1e-3f, 2e-3f}, z = {0.999f, 1.001f};
  float fx = 1.0f + tid;
  uint32_t pk = 0;
  const float fa = 1.0f + (tid & 3), fb = 0.5f;
  for (int i = 0; i < iters; ++i) {
    if (LDS_FED) {
      a = __builtin_bit_cast(bf16x8, lds[(tid + i) & 255]);
      b = __builtin_bit_cast(bf16x8, lds[256 + ((tid + 3 * i) & 255)]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x16& c = m == 0 ? c0 : m == 1 ? c1 : m == 2 ? c2 : c3;
      if (BF16) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (VALU == V_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        if (VALU == V_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(z), "v"(y));
        if (VALU == V_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(z));
        if (VALU == V_CVT_PK) { uint32_t t; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(fx), "v"(fb)); pk += t; }
        if (VALU == V_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(fx) : "v"(fb), "v"(fa));
        if (VALU == V_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(pk) : "v"(fx));
        if (VALU == V_SNOP) asm volatile("s_nop 7");                        // 8 idle issue cycles, no instruction at all
      }
    }
    if (BARRIER) __syncthreads();
  }
  const float r = c0[0] + c1[5] + c2[9] + c3[15] + x[0] + x[1] + fx + static_cast<float>(pk & 255u);
  if (r == 12345.678f) sink[tid] = r;
}

template <bool BF16, int VALU, int NV, bool LDS_FED, bool BARRIER>
static void launch_combo(float* sink, int grid, int iters, hipStream_t st) {
  hipLaunchKernelGGL((combo_offender_kernel<BF16, VALU, NV, LDS_FED, BARRIER>), dim3(grid), dim3(256), 8192, st, sink, iters);
  CHECK(hipGetLastError());
}

// Which matrix instructions do it?  Four MFMAs of one flavour + six v_fmac_f32 per loop trip (the strongest mix above).
enum { F_F32_32x32x2, F_BF16_32x32x16, F_BF16_32x32x8_1K, F_BF16_16x16x32, F_F16_32x32x16, F_F16_32x32x8, F_F32_16x16x4,
       F_I8_32x32x32, F_F64_16x16x4, F_BF16_16x16x16_1K, F_COUNT };
static const char* kFlavName[F_COUNT] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x8_bf16_1k",
    "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_16x16x4_f32",
    "v_mfma_i32_32x32x32_i8", "v_mfma_f64_16x16x4_f64", "v_mfma_f32_16x16x16_bf16_1k"};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int FLAV, int NV>
__global__ __launch_bounds__(256) void flavour_offender_kernel(float* sink, int iters) {
  const int tid = threadIdx.x;
  float fx = 1.0f + tid;
  const float fa = 1.0f + (tid & 3), fb = 0.5f;
  float r = 0.f;
#define VALU_FILL() _Pragma("unroll") for (int v = 0; v < NV; ++v) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(fx) : "v"(fb), "v"(fa))
  if (FLAV == F_F32_32x32x2) {
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c[m], 0, 0, 0); VALU_FILL(); }
    r = c[0][0] + c[1][5] + c[2][9] + c[3][15];
  } else if (FLAV == F_F32_16x16x4) {
    f32x4 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c[m], 0, 0, 0); VALU_FILL(); }
    r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
  } else if (FLAV == F_BF16_32x32x16 || FLAV == F_BF16_16x16x32) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = static_cast<short>(0x3f80 + ((tid + k) & 7)); b[k] = static_cast<short>(0x3f80 + ((tid * 3 + k) & 7)); }
    if (FLAV == F_BF16_32x32x16) {
      f32x16 c[4] = {{0}, {0}, {0}, {0}};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[m], 0, 0, 0); VALU_FILL(); }
      r = c[0][0] + c[1][5] + c[2][9] + c[3][15];
    } else {
      f32x4 c[4] = {{0}, {0}, {0}, {0}};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[m], 0, 0, 0); VALU_FILL(); }
      r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
  } else if (FLAV == F_BF16_32x32x8_1K || FLAV == F_BF16_16x16x16_1K) {
    s16x4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = static_cast<short>(0x3f80 + ((tid + k) & 7)); b[k] = static_cast<short>(0x3f80 + ((tid * 3 + k) & 7)); }
    if (FLAV == F_BF16_32x32x8_1K) {
      f32x16 c[4] = {{0}, {0}, {0}, {0}};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c[m], 0, 0, 0); VALU_FILL(); }
      r = c[0][0] + c[1][5] + c[2][9] + c[3][15];
    } else {
      f32x4 c[4] = {{0}, {0}, {0}, {0}};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c[m], 0, 0, 0); VALU_FILL(); }
      r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
  } else if (FLAV == F_F16_32x32x16) {
    h16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = static_cast<_Float16>(1 + ((tid + k) & 3)); b[k] = static_cast<_Float16>(0.5f); }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[m], 0, 0, 0); VALU_FILL(); }
    r = c[0][0] + c[1][5] + c[2][9] + c[3][15];
  } else if (FLAV == F_F16_32x32x8) {
    h16x4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = static_cast<_Float16>(1 + ((tid + k) & 3)); b[k] = static_cast<_Float16>(0.5f); }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c[m], 0, 0, 0); VALU_FILL(); }
    r = c[0][0] + c[1][5] + c[2][9] + c[3][15];
  } else if (FLAV == F_I8_32x32x32) {
    i32x4 a = {tid, 1, 2, 3}, b = {1, 1, 1, 1};
    i32x16 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[m], 0, 0, 0); VALU_FILL(); }
    r = static_cast<float>(c[0][0] + c[1][5] + c[2][9] + c[3][15]);
  } else if (FLAV == F_F64_16x16x4) {
    f64x4 c[4] = {{0}, {0}, {0}, {0}};
    const double da = 1.0 + (tid & 3), db = 0.5;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) { c[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, c[m], 0, 0, 0); VALU_FILL(); }
    r = static_cast<float>(c[0][0] + c[1][1] + c[2][2] + c[3][3]);
  }
#undef VALU_FILL
  r += fx;
  if (r == 12345.678f) sink[tid] = r;
}

template <int FLAV, int NV>
static void launch_flavour(float* sink, int grid, int iters, hipStream_t st) {
  hipLaunchKernelGGL((flavour_offender_kernel<FLAV, NV>), dim3(grid), dim3(256), 0, st, sink, iters);
  CHECK(hipGetLastError());
}

template <int BF3>
static void launch_tile(float* sink, const float* a_rows, const float* b_rows, int grid, int iters, hipStream_t st) {
  constexpr int bytes = static_cast<int>(cmhse::TileSmem<128, 256>::kBytes);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_offender_kernel<BF3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  hipLaunchKernelGGL(tile_offender_kernel<BF3>, dim3(grid), dim3(256), bytes, st, sink, a_rows, b_rows, 1024, iters);
  CHECK(hipGetLastError());
}

template <int MODE, int PAD>
static void launch_offender(float* sink, const float* src, int grid, int lds_bytes, int iters, hipStream_t st) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(offender_kernel<MODE, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipLaunchKernelGGL((offender_kernel<MODE, PAD>), dim3(grid), dim3(256), lds_bytes, st, sink, src, iters);
  CHECK(hipGetLastError());
}

template <int PAD>
static void launch_mode(int mode, float* sink, const float* src, int grid, int lds_bytes, int iters, hipStream_t st) {
  switch (mode) {
    case M_IDLE: launch_offender<M_IDLE, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_MFMA_BF16: launch_offender<M_MFMA_BF16, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_MFMA_F32: launch_offender<M_MFMA_F32, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_CVT_PK: launch_offender<M_CVT_PK, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_PK_ADD: launch_offender<M_PK_ADD, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_LDS: launch_offender<M_LDS, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_PK_FMA: launch_offender<M_PK_FMA, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_MFMA_BF16_16: launch_offender<M_MFMA_BF16_16, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_VALU_F32: launch_offender<M_VALU_F32, PAD>(sink, src, grid, lds_bytes, iters, st); break;
    case M_GLOBAL: launch_offender<M_GLOBAL, PAD>(sink, src, grid, lds_bytes, iters, st); break;
  }
}

static void launch_pad(int pad, int mode, float* sink, const float* src, int grid, int lds_bytes, int iters, hipStream_t st) {
  switch (pad) {
    case 0: launch_mode<0>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 200: launch_mode<200>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 224: launch_mode<224>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 232: launch_mode<232>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 240: launch_mode<240>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 248: launch_mode<248>(mode, sink, src, grid, lds_bytes, iters, st); break;
    case 256: launch_mode<256>(mode, sink, src, grid, lds_bytes, iters, st); break;
    default: printf("no such pad\n"); exit(2);
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 4;
  const int rows = 24576, len = 80, blocks = 4096, viters = 24;
  float *hs, *sink;
  uint32_t* report;
  CHECK(hipMalloc(&hs, static_cast<size_t>(rows) * 1024 * 4));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMalloc(&report, 1024));
  hipStream_t s_v, s_o;
  CHECK(hipStreamCreateWithFlags(&s_v, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s_o, hipStreamNonBlocking));
  hipLaunchKernelGGL(fill_rows_pattern, dim3(rows * 1024 / 256), dim3(256), 0, s_v, hs, static_cast<uint32_t>(rows) * 1024u);
  CHECK(hipStreamSynchronize(s_v));
  hipEvent_t e0, e1, o0, o1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&o0)); CHECK(hipEventCreate(&o1));
  // offender iteration counts sized to last about as long as the victim launch (~10-20 ms)
  const int oiters[M_COUNT] = {400000, 150000, 40000, 300000, 300000, 60000, 300000, 300000, 300000, 40000};
  float* wbuf;
  CHECK(hipMalloc(&wbuf, 256 * 1024 * 4));
  hipLaunchKernelGGL(fill_rows_pattern, dim3(256 * 1024 / 256), dim3(256), 0, s_v, wbuf, 256u * 1024u);
  CHECK(hipStreamSynchronize(s_v));
  const int tgrid = argc > 3 ? atoi(argv[3]) : 64;
  for (int packed = 1; packed >= 0; --packed)
    for (int bf3 = 2; bf3 >= 0; --bf3) {
      CHECK(hipMemset(report, 0, 1024));
      float vms = 0.f, oms = 0.f;
      for (int rep = 0; rep < 4 * reps; ++rep) {
        CHECK(hipEventRecord(o0, s_o));
        if (bf3 == 2) launch_tile<2>(sink, hs, wbuf, tgrid, 400, s_o); else if (bf3) launch_tile<1>(sink, hs, wbuf, tgrid, 400, s_o); else launch_tile<0>(sink, hs, wbuf, tgrid, 100, s_o);
        CHECK(hipEventRecord(o1, s_o));
        CHECK(hipEventRecord(e0, s_v));
        if (packed)
          hipLaunchKernelGGL(victim_kernel<VK_PK_FMA>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
        else
          hipLaunchKernelGGL(victim_kernel<VK_FMAC>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
        CHECK(hipEventRecord(e1, s_v));
        CHECK(hipDeviceSynchronize());
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, o0, o1));
        vms += a; oms += b;
      }
      uint32_t h[64];
      CHECK(hipMemcpy(h, report, 256, hipMemcpyDeviceToHost));
      printf("victim %-12s beside the product's tile loop %-28s (%d workgroups): %8u wrong sums of %.2e  (victim %.1f ms, offender %.1f ms; lane quarters hit 0x%x)\n",
             packed ? "v_pk_fma_f32" : "v_fmac_f32", bf3 == 2 ? "nt_phase_bf3, A pre-split" : bf3 ? "nt_phase_bf3 (bf16x3)" : "nt_phase (fp32)", tgrid, h[1],
             static_cast<double>(h[0]) * 1024.0, vms / (4 * reps), oms / (4 * reps), h[3]);
      for (uint32_t i = 0; i < (h[2] < 2 ? h[2] : 2); ++i) {
        const uint32_t* q = h + 8 + 8 * i;
        float got;
        memcpy(&got, &q[4], 4);
        printf("      block %u iteration %u thread %u (lane %u) component %u: got %.1f want %u\n", q[0], q[1], q[2], q[2] & 63, q[3], got, q[5]);
      }
    }
  {
    struct Arm { const char* name; void (*launch)(float*, int, int, hipStream_t); int iters; };
    const Arm arms[] = {
        {"bf16 MFMA alone", launch_combo<true, V_NONE, 0, false, false>, 150000},
        {"bf16 MFMA, LDS-fed operands", launch_combo<true, V_NONE, 0, true, false>, 150000},
        {"bf16 MFMA, LDS-fed, barrier", launch_combo<true, V_NONE, 0, true, true>, 150000},
        {"bf16 MFMA + 2 v_pk_add_f32", launch_combo<true, V_PK_ADD, 2, false, false>, 150000},
        {"bf16 MFMA + 2 v_pk_fma_f32", launch_combo<true, V_PK_FMA, 2, false, false>, 150000},
        {"bf16 MFMA + 2 v_pk_mul_f32", launch_combo<true, V_PK_MUL, 2, false, false>, 150000},
        {"bf16 MFMA + 2 v_cvt_pk_bf16_f32", launch_combo<true, V_CVT_PK, 2, false, false>, 150000},
        {"bf16 MFMA + 2 v_fmac_f32", launch_combo<true, V_FMAC, 2, false, false>, 150000},
        {"bf16 MFMA + 2 v_mov_b32", launch_combo<true, V_MOV, 2, false, false>, 150000},
        {"bf16 MFMA + 6 v_fmac_f32", launch_combo<true, V_FMAC, 6, false, false>, 100000},
        {"bf16 MFMA + 6 v_pk_add_f32", launch_combo<true, V_PK_ADD, 6, false, false>, 100000},
        {"bf16 MFMA + 6 s_nop 7 (idle gaps only)", launch_combo<true, V_SNOP, 6, false, false>, 100000},
        {"bf16 MFMA + 2 s_nop 7", launch_combo<true, V_SNOP, 2, false, false>, 100000},
        {"fp32 MFMA + 2 v_pk_add_f32", launch_combo<false, V_PK_ADD, 2, false, false>, 40000},
        {"fp32 MFMA + 2 v_fmac_f32", launch_combo<false, V_FMAC, 2, false, false>, 40000},
        {"fp32 MFMA + 6 v_pk_add_f32", launch_combo<false, V_PK_ADD, 6, false, false>, 40000},
    };
    const int cgrid = argc > 2 ? atoi(argv[2]) : 256;
    for (int packed = 1; packed >= 0; --packed)
      for (const Arm& arm : arms) {
        CHECK(hipMemset(report, 0, 1024));
        float vms = 0.f, oms = 0.f;
        for (int rep = 0; rep < reps; ++rep) {
          CHECK(hipEventRecord(o0, s_o));
          arm.launch(sink, cgrid, arm.iters, s_o);
          CHECK(hipEventRecord(o1, s_o));
          CHECK(hipEventRecord(e0, s_v));
          if (packed)
            hipLaunchKernelGGL(victim_kernel<VK_PK_FMA>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
          else
            hipLaunchKernelGGL(victim_kernel<VK_FMAC>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
          CHECK(hipEventRecord(e1, s_v));
          CHECK(hipDeviceSynchronize());
          float a, b;
          CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, o0, o1));
          vms += a; oms += b;
        }
        uint32_t h[64];
        CHECK(hipMemcpy(h, report, 256, hipMemcpyDeviceToHost));
        printf("victim %-12s beside %-34s (%d workgroups): %8u wrong sums of %.2e  (victim %.1f ms, offender %.1f ms; lane quarters hit 0x%x)\n",
               packed ? "v_pk_fma_f32" : "v_fmac_f32", arm.name, cgrid, h[1], static_cast<double>(h[0]) * 1024.0, vms / reps, oms / reps, h[3]);
      }
  }
  {   // which instructions of a bystander lose updates?  beside the tile loop and beside the strongest synthetic mix
    for (int off = 0; off < 2; ++off)
      for (int vk = 0; vk < VK_COUNT; ++vk) {
        CHECK(hipMemset(report, 0, 1024));
        for (int rep = 0; rep < 2 * reps; ++rep) {
          if (off == 0) launch_tile<1>(sink, hs, wbuf, tgrid, 400, s_o);
          else launch_combo<true, V_FMAC, 6, false, false>(sink, 256, 100000, s_o);
          switch (vk) {
            case VK_FMAC: hipLaunchKernelGGL(victim_kernel<VK_FMAC>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters); break;
            case VK_PK_FMA: hipLaunchKernelGGL(victim_kernel<VK_PK_FMA>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters); break;
            case VK_PK_MUL_ADD: hipLaunchKernelGGL(victim_kernel<VK_PK_MUL_ADD>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters); break;
            case VK_FMA_F64: hipLaunchKernelGGL(victim_kernel<VK_FMA_F64>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters); break;
            case VK_PK_ADD: hipLaunchKernelGGL(victim_kernel<VK_PK_ADD>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters); break;
          }
          CHECK(hipDeviceSynchronize());
        }
        uint32_t h[64];
        CHECK(hipMemcpy(h, report, 256, hipMemcpyDeviceToHost));
        printf("victim accumulating with %-28s beside %-40s: %8u wrong sums of %.2e  (lane quarters hit 0x%x)\n", kVictimName[vk],
               off == 0 ? "the tile loop nt_phase_bf3" : "4 x v_mfma_f32_32x32x16_bf16 + 24 v_fmac_f32", h[1], static_cast<double>(h[0]) * 1024.0, h[3]);
      }
  }
  {
    struct Arm { int flav; int nv; void (*launch)(float*, int, int, hipStream_t); int iters; };
#define FARM(f, it) {f, 6, launch_flavour<f, 6>, it}, {f, 12, launch_flavour<f, 12>, it}
    const Arm arms[] = {FARM(F_F32_32x32x2, 40000), FARM(F_F32_16x16x4, 150000), FARM(F_BF16_32x32x16, 100000),
                        FARM(F_BF16_32x32x8_1K, 100000), FARM(F_BF16_16x16x32, 150000), FARM(F_BF16_16x16x16_1K, 150000),
                        FARM(F_F16_32x32x16, 100000), FARM(F_F16_32x32x8, 100000), FARM(F_I8_32x32x32, 100000), FARM(F_F64_16x16x4, 80000)};
#undef FARM
    const int cgrid = argc > 2 ? atoi(argv[2]) : 256;
    for (const Arm& arm : arms) {
      CHECK(hipMemset(report, 0, 1024));
      float vms = 0.f, oms = 0.f;
      for (int rep = 0; rep < reps; ++rep) {
        CHECK(hipEventRecord(o0, s_o));
        arm.launch(sink, cgrid, arm.iters, s_o);
        CHECK(hipEventRecord(o1, s_o));
        CHECK(hipEventRecord(e0, s_v));
        hipLaunchKernelGGL(victim_kernel<VK_PK_FMA>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
        CHECK(hipEventRecord(e1, s_v));
        CHECK(hipDeviceSynchronize());
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, o0, o1));
        vms += a; oms += b;
      }
      uint32_t h[64];
      CHECK(hipMemcpy(h, report, 256, hipMemcpyDeviceToHost));
      printf("victim v_pk_fma_f32 beside 4 x %-28s + %2d v_fmac_f32 per MFMA (%d workgroups): %8u wrong sums of %.2e  (victim %.1f ms, offender %.1f ms; lane quarters hit 0x%x)\n",
             kFlavName[arm.flav], arm.nv, cgrid, h[1], static_cast<double>(h[0]) * 1024.0, vms / reps, oms / reps, h[3]);
    }
  }
  if (argc > 4) return 0;   // section 5 (long, all clean) only on request
  const int pads[] = {0, 200, 224, 232, 240, 248, 256};
  const int modes[] = {M_IDLE, M_MFMA_BF16, M_MFMA_F32, M_PK_FMA, M_LDS};
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  for (int packed = 1; packed >= 0; --packed) {
    printf("victim: %s accumulators; offender grid %d workgroups\n", packed ? "v_pk_fma_f32" : "v_fmac_f32", grid);
    for (int mi = 0; mi < 5; ++mi) {
      const int mode = modes[mi];
      for (int pi = 0; pi < 7; ++pi) {
        const int pad = pads[pi], lds_kb = 60;
        CHECK(hipMemset(report, 0, 1024));
        float vms = 0.f, oms = 0.f;
        for (int rep = 0; rep < reps; ++rep) {
          CHECK(hipEventRecord(o0, s_o));
          launch_pad(pad, mode, sink, hs, grid, lds_kb * 1024, oiters[mode], s_o);
          CHECK(hipEventRecord(o1, s_o));
          CHECK(hipEventRecord(e0, s_v));
          if (packed)
            hipLaunchKernelGGL(victim_kernel<VK_PK_FMA>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
          else
            hipLaunchKernelGGL(victim_kernel<VK_FMAC>, dim3(blocks), dim3(256), 0, s_v, report, hs, rows, len, viters);
          CHECK(hipEventRecord(e1, s_v));
          CHECK(hipDeviceSynchronize());
          float a, b;
          CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, o0, o1));
          vms += a; oms += b;
        }
        uint32_t h[64];
        CHECK(hipMemcpy(h, report, 256, hipMemcpyDeviceToHost));
        printf("  offender %-28s VGPRs %3d: %8u wrong sums of %.2e  (victim %.1f ms, offender %.1f ms; lane quarters hit 0x%x)\n",
               kModeName[mode], pad, h[1], static_cast<double>(h[0]) * 1024.0, vms / reps, oms / reps, h[3]);
        for (uint32_t i = 0; i < (h[2] < 2 ? h[2] : 2); ++i) {
          const uint32_t* q = h + 8 + 8 * i;
          float got;
          memcpy(&got, &q[4], 4);
          printf("      block %u iteration %u thread %u (lane %u) component %u: got %.1f want %u\n", q[0], q[1], q[2], q[2] & 63, q[3], got, q[5]);
        }
      }
    }
  }
  return 0;
}
