// Where do the recurrent weights of a small-batch GRU step come from?  Every step kernel re-reads
// W_hh (12.6 MB fp32 at H = 1024), each workgroup its own slice.  This measures the rate at which a
// workgroup streams its slice (a) on the first pass after a kernel boundary and (b) on a second
// pass inside the same kernel (slice then resident in the XCD's L2), for 64 and 256 workgroups —
// i.e. what a kernel that stays resident over the steps of a chain would gain on the operand side.
//   hipcc --offload-arch=gfx950 -O3 -o weights_reread weights_reread.hip && ./weights_reread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kPasses = 4;

__global__ __launch_bounds__(256) void stream_slice(const float4* __restrict__ w, size_t slice_f4,
                                                    long long* stamps, float* sink) {
  const float4* base = w + static_cast<size_t>(blockIdx.x) * slice_f4;
  float acc = 0.f;
  if (threadIdx.x == 0) stamps[blockIdx.x * (kPasses + 1)] = wall_clock64();
  for (int pass = 0; pass < kPasses; ++pass) {
    // 8 x 16-byte loads in flight per lane
    for (size_t i = threadIdx.x; i < slice_f4; i += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t k = i + 256 * j;
        v[j] = (k < slice_f4) ? base[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x * (kPasses + 1) + pass + 1] = wall_clock64();
  }
  if (acc == 123.456f) sink[0] = acc;
}

// The same bytes through the step kernel's MFMA-operand pattern: lane (r = lane & 15, kq = lane >> 4)
// loads 16 bytes of row r (row stride 4 KB) at k = 16 blk + 4 kq; the 4 waves interleave pairs of
// 16-k blocks; 3 gate rows per block; 4 blocks (12 loads) in flight per lane.
__global__ __launch_bounds__(256) void stream_rows(const float* __restrict__ w, int H, long long* stamps,
                                                   float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * 16;
  float acc = 0.f;
  if (threadIdx.x == 0) stamps[blockIdx.x * (kPasses + 1)] = wall_clock64();
  for (int pass = 0; pass < kPasses; ++pass) {
    for (int p = 0; p < H / 16 / 8; p += 2) {      // pairs of blocks: wave's pair index p*4 + wave
      float4 v[12];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const int blk = 2 * ((p + pp) * 4 + wave) + half;
            v[(pp * 2 + half) * 3 + g] = *reinterpret_cast<const float4*>(
                w + (static_cast<size_t>(g) * H + u0 + r) * H + blk * 16 + 4 * kq);
          }
#pragma unroll
      for (int j = 0; j < 12; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x * (kPasses + 1) + pass + 1] = wall_clock64();
  }
  if (acc == 123.456f) sink[0] = acc;
}

__global__ void dirty(float* p, size_t n) {   // what a step's own stores do between two steps
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = 1.f;
}

int main() {
  const size_t total_f4 = 3072ull * 1024 / 4;   // W_hh [3H, H] at H = 1024
  float4* w; long long* stamps; float* sink; float* scratch;
  CK(hipMalloc(reinterpret_cast<void**>(&w), total_f4 * 16));
  CK(hipMemset(w, 0, total_f4 * 16));
  CK(hipMalloc(reinterpret_cast<void**>(&stamps), 1024 * (kPasses + 1) * 8));
  CK(hipMalloc(reinterpret_cast<void**>(&sink), 4));
  CK(hipMalloc(reinterpret_cast<void**>(&scratch), 1 << 20));
  std::vector<long long> h(1024 * (kPasses + 1));
  for (int n_wg : {64, 128, 256}) {
    const size_t slice = total_f4 / n_wg;
    for (int rep = 0; rep < 4; ++rep) {
      if (rep >= 2) hipLaunchKernelGGL(dirty, dim3(64), dim3(256), 0, 0, scratch, (1u << 20) / 4);
      hipLaunchKernelGGL(stream_slice, dim3(n_wg), dim3(256), 0, 0, w, slice, stamps, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), stamps, n_wg * (kPasses + 1) * 8, hipMemcpyDeviceToHost));
      printf("%3d workgroups x %4zu KB, launch %d%s:", n_wg, slice * 16 / 1024, rep,
             rep >= 2 ? " (after a storing kernel)" : "");
      for (int p = 0; p < kPasses; ++p) {
        std::vector<double> us;
        for (int g = 0; g < n_wg; ++g)
          us.push_back((h[g * (kPasses + 1) + p + 1] - h[g * (kPasses + 1) + p]) * 0.01);
        std::sort(us.begin(), us.end());
        const double med = us[us.size() / 2];
        printf("  pass %d %6.2f us (%5.1f GB/s per WG, max %6.2f)", p, med, slice * 16 / med * 1e-3, us.back());
      }
      printf("\n");
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_rows, dim3(64), dim3(256), 0, 0, reinterpret_cast<const float*>(w), 1024, stamps, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), stamps, 64 * (kPasses + 1) * 8, hipMemcpyDeviceToHost));
    printf(" 64 workgroups x 48 rows x 4 KB in MFMA-operand order, launch %d:", rep);
    for (int p = 0; p < kPasses; ++p) {
      std::vector<double> us;
      for (int g = 0; g < 64; ++g) us.push_back((h[g * (kPasses + 1) + p + 1] - h[g * (kPasses + 1) + p]) * 0.01);
      std::sort(us.begin(), us.end());
      printf("  pass %d %6.2f us (%5.1f GB/s per WG)", p, us[32], 196608 / us[32] * 1e-3);
    }
    printf("\n");
  }
  return 0;
}
