// Bare fp32 MFMA loops on random operands held in registers: does the chip sustain a different
// clock (and FLOP rate) on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32?
// (MI355X_MICROARCH.md "DVFS give-back" item 7 reports 1.12-1.15x for the bf16 pair of shapes.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_f32 mfma_shape_f32.hip && ./mfma_shape_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 4096;

// per wave: 32 x 128 output tile = 4 accumulators of 32x32 (64 registers); per k-pair 4 MFMAs
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void loop_32x32x2(const float* __restrict__ in, float* out,
                                                    unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  float a[4], b[4][4];
  for (int j = 0; j < 4; ++j) {
    a[j] = in[(threadIdx.x * 20 + j) & 4095];
    for (int n = 0; n < 4; ++n) b[n][j] = in[(threadIdx.x * 20 + 4 + n * 4 + j) & 4095];
  }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[n][j], acc[n], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = c1 - c0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
  (void)lane;
}

// same output tile per wave: 32 x 128 = 2 x 8 blocks of 16x16 (64 registers); per 4 k: 16 MFMAs
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void loop_16x16x4(const float* __restrict__ in, float* out,
                                                    unsigned long long* stamps) {
  float a[2][2], b[8][2];
  for (int j = 0; j < 2; ++j) {
    for (int m = 0; m < 2; ++m) a[m][j] = in[(threadIdx.x * 20 + m * 2 + j) & 4095];
    for (int n = 0; n < 8; ++n) b[n][j] = in[(threadIdx.x * 20 + 4 + n * 2 + j) & 4095];
  }
  f32x4 acc[2][8];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 8; ++n)
      for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
  // one iteration = 8 k (2 steps of 4 k) = the same FLOPs as 4 k-pairs of the 32x32x2 loop
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = wall_clock64();
  float s = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 8; ++n)
      for (int i = 0; i < 4; ++i) s += acc[m][n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = c1 - c0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <typename K>
static void run(const char* name, K kernel, int blocks, const float* d_in, float* d_out,
                unsigned long long* d_st) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_st);
  hipDeviceSynchronize();
  const int reps = 200;   // ~ seconds of back-to-back launches so DVFS settles
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_st);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st(blocks * 2);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int b = 0; b < blocks; ++b) {
    cyc += st[b * 2];
    rt += st[b * 2 + 1];
  }
  // FLOPs per wave per iteration: 4 k-pairs x 4 MFMAs x (32*32*2*2)
  const double flops = double(blocks) * 4 * kIters * 16.0 * 4096.0 * reps;
  printf("%-14s blocks %5d  %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz  cycles/iter/wave %.1f\n",
         name, blocks, ms, flops / (ms * 1e-3) / 1e12, cyc / (rt * 10.0), cyc / blocks / kIters);
}

int main() {
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / float(RAND_MAX) - 0.5f) * 2.0f;
  float *d_in, *d_out;
  unsigned long long* d_st;
  hipMalloc(&d_in, 4096 * 4);
  hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int max_blocks = 256 * 3;
  hipMalloc(&d_out, max_blocks * 256 * 4);
  hipMalloc(&d_st, max_blocks * 16);
  for (int rep = 0; rep < 2; ++rep) {
    for (int wps = 1; wps <= 3; ++wps) {
      const int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD
      char n1[32], n2[32];
      snprintf(n1, sizeof n1, "32x32x2  w%d", wps);
      snprintf(n2, sizeof n2, "16x16x4  w%d", wps);
      run(n1, loop_32x32x2<1>, blocks, d_in, d_out, d_st);
      run(n2, loop_16x16x4<1>, blocks, d_in, d_out, d_st);
    }
  }
  // all-zero operands: the rate both shapes reach when the clock is not held down
  hipMemset(d_in, 0, 4096 * 4);
  run("32x32x2 zero", loop_32x32x2<1>, 256, d_in, d_out, d_st);
  run("16x16x4 zero", loop_16x16x4<1>, 256, d_in, d_out, d_st);
  return 0;
}
