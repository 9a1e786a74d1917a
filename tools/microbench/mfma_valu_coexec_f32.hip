// Can exact-fp32 throughput be raised above the fp32 MFMA peak by running v_fma_f32 / v_pk_fma_f32 waves
// BESIDE the v_mfma_f32_32x32x2_f32 waves on the same SIMDs?  (The two are separate pipes for bf16
// MFMA; for fp32 the matrix rate equals the vector rate, 64 FLOP/clk/SIMD, which smells of shared
// multipliers.)  512-thread workgroups, one per CU: waves 0-3 run a bare MFMA loop, waves 4-7 a bare
// VALU FMA loop (64 independent accumulators per lane), operands in registers, no memory traffic.
// Three launches: MFMA waves only, VALU waves only, both; work per wave identical in all three.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coexec_f32.bin mfma_valu_coexec_f32.hip && ./mfma_valu_coexec_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 8192;

// mode bit 0: MFMA waves work; bit 1: VALU waves work; PK: v_pk_fma_f32 instead of v_fma_f32
template <bool PK>
__global__ __launch_bounds__(512) void coexec(const float* __restrict__ in, float* out, int mode,
                                              unsigned long long* stamps) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long r0 = wall_clock64();
  float s = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      float a[4], b[4][4];
      for (int j = 0; j < 4; ++j) {
        a[j] = in[(threadIdx.x * 20 + j) & 4095];
        for (int n = 0; n < 4; ++n) b[n][j] = in[(threadIdx.x * 20 + 4 + n * 4 + j) & 4095];
      }
      f32x16 acc[4];
      for (int n = 0; n < 4; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
      for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[n][j], acc[n], 0, 0, 0);
      }
      for (int n = 0; n < 4; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    }
  } else if (mode & 2) {
    // 16 MFMAs of 32x32x2 = 16 x 4096 FMAs per wave = 1024 per lane: the same FLOPs per iteration here
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) {
      a[j] = in[(threadIdx.x * 24 + j) & 4095];
      b[j] = in[(threadIdx.x * 24 + 8 + j) & 4095];
    }
    if (PK) {
      f32x2 acc[32];
      for (int i = 0; i < 32; ++i) acc[i] = f32x2{0.f, 0.f};
      for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)        // 16 x 32 pk_fma = 1024 FMAs per lane
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            const f32x2 av = {a[(i + r) & 7], a[(i + r + 1) & 7]}, bv = {b[i & 7], b[(i + 3) & 7]};
            acc[i] = __builtin_elementwise_fma(av, bv, acc[i]);
          }
      }
      for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y;
    } else {
      float acc[64];
      for (int i = 0; i < 64; ++i) acc[i] = 0.f;
      for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)        // 16 x 64 v_fma = 1024 FMAs per lane
#pragma unroll
          for (int i = 0; i < 64; ++i) acc[i] = __builtin_fmaf(a[(i + r) & 7], b[i & 7], acc[i]);
      }
      for (int i = 0; i < 64; ++i) s += acc[i];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  const unsigned long long r1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = r1 - r0;
}

int main() {
  const int grid = 256;
  float *in, *out;
  unsigned long long* st;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, grid * 512 * 4);
  hipMalloc(&st, grid * 8 * 8);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (i % 17 - 8) * 1e-3f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const double flop_wave = double(kIters) * 16 * 2.0 * 32 * 32 * 2;   // per wave per launch, either kind
  for (int pk = 0; pk < 2; ++pk)
    for (int mode = 1; mode <= 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(coexec<true>, dim3(grid), dim3(512), 0, 0, in, out, mode, st);
        else hipLaunchKernelGGL(coexec<false>, dim3(grid), dim3(512), 0, 0, in, out, mode, st);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const int kinds = (mode == 3) ? 2 : 1;
      const double tf = flop_wave * 4 * kinds * grid / (ms * 1e-3) / 1e12;
      printf("%-12s mode %d (%s): %.3f ms  %.1f TFLOP/s total (fp32 peak of either pipe: 157.3)\n", pk ? "v_pk_fma_f32" : "v_fma_f32",
             mode, mode == 1 ? "MFMA waves only" : (mode == 2 ? "VALU waves only" : "both"), ms, tf);
    }
  return 0;
}
