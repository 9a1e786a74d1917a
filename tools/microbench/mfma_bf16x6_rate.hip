// Kill criterion of the 3-way bf16 split ("bf16x6": a = ah + am + al, 8 + 8 + 8 mantissa bits; the six
// products hh, hm, mh, hl, lh, mm on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; error ~2^-24 per
// product): at the clock the chip holds, how much faster is the SAME contraction volume
// (a 32 x 128 output tile per wave, K = 16 per iteration) on six bf16 MFMAs than on eight fp32 MFMAs?
// Bare loops, operands in registers; 256-thread workgroups, 2 or 3 per CU like the step tile.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16x6_rate.bin mfma_bf16x6_rate.hip && ./mfma_bf16x6_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 4096;

__global__ __launch_bounds__(256) void loop_fp32(const float* __restrict__ in, float* out) {
  float a[8], b[4][8];
  for (int j = 0; j < 8; ++j) {
    a[j] = in[(threadIdx.x * 40 + j) & 4095];
    for (int n = 0; n < 4; ++n) b[n][j] = in[(threadIdx.x * 40 + 8 + n * 8 + j) & 4095];
  }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)          // K = 16: eight 32x32x2 steps
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[n][j], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PRODUCTS>
__global__ __launch_bounds__(256) void loop_bf16(const float* __restrict__ in, float* out) {
  bf16x8 a[3], b[4][3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) {
      a[p][j] = static_cast<__bf16>(in[(threadIdx.x * 40 + p * 8 + j) & 4095]);
      for (int n = 0; n < 4; ++n) b[n][p][j] = static_cast<__bf16>(in[(threadIdx.x * 40 + 24 + n * 8 + p + j) & 4095]);
    }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  // (piece of a, piece of b) per product: hh, hm, mh, hl, lh, mm; the 3-product form stops after mh
  const int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int q = 0; q < PRODUCTS; ++q)   // K = 16 in ONE 32x32x16 step per product
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[q]], b[n][pb[q]], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096 * 4);
  (void)hipMalloc(&out, 768 * 256 * 4);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (i % 17 - 8) * 1e-3f;
  (void)hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int per_cu = 2; per_cu <= 3; ++per_cu) {
    const int grid = 256 * per_cu;
    const double flop = double(kIters) * 4 * 2.0 * 32 * 32 * 16 * 4 * grid;   // fp32-equivalent products
    float ms[3];
    for (int k = 0; k < 3; ++k) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (k == 0) hipLaunchKernelGGL(loop_fp32, dim3(grid), dim3(256), 0, 0, in, out);
        else if (k == 1) hipLaunchKernelGGL(loop_bf16<3>, dim3(grid), dim3(256), 0, 0, in, out);
        else hipLaunchKernelGGL(loop_bf16<6>, dim3(grid), dim3(256), 0, 0, in, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      (void)hipEventElapsedTime(&ms[k], e0, e1);
    }
    printf("%d workgroups per CU: fp32 32x32x2 %.3f ms = %.1f TFLOP/s | bf16x3 %.3f ms = %.1f (%.2fx) | bf16x6 %.3f ms = %.1f (%.2fx)\n",
           per_cu, ms[0], flop / ms[0] / 1e9, ms[1], flop / ms[1] / 1e9, ms[0] / ms[1], ms[2], flop / ms[2] / 1e9, ms[0] / ms[2]);
  }
  return 0;
}
