// How fast can the [S,T,I] loader tensors of a validation split travel host -> HBM in TIME CHUNKS
// (all sequences x steps [t0,t1)), the unit the step pipeline consumes?
//   A  one hipMemcpyAsync per whole tensor (contiguous; the upper bound)
//   B  hipMemcpy2DAsync per (tensor, chunk): rows of (t1-t0)*I*4 bytes, pitch T*I*4
//   C  one "pull" kernel per chunk reading the pinned host memory directly (zero-copy over PCIe)
//   hipcc --offload-arch=gfx950 -O3 -o h2d_chunked h2d_chunked.hip && ./h2d_chunked
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Desc { const float* src; float* dst; int S; };

// one chunk of every tensor: row r of tensor d, bytes [t0*I*4, t1*I*4) of its T*I*4-byte row
__global__ __launch_bounds__(256) void pull_chunk(const Desc* descs, int n_desc, int T, int I, int t0, int t1) {
  const int d = blockIdx.y;
  const Desc dd = descs[d];
  const size_t row_f4 = static_cast<size_t>(T) * I / 4;          // float4 per row
  const size_t w_f4 = static_cast<size_t>(t1 - t0) * I / 4;      // float4 per row chunk
  const size_t off = static_cast<size_t>(t0) * I / 4;
  const float4* s = reinterpret_cast<const float4*>(dd.src);
  float4* o = reinterpret_cast<float4*>(dd.dst);
  const size_t total = static_cast<size_t>(dd.S) * w_f4;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += gridDim.x * 256ull) {
    const size_t r = i / w_f4, c = i % w_f4;
    o[r * row_f4 + off + c] = s[r * row_f4 + off + c];
  }
}

int main() {
  const int S = 120, T = 80, I = 2048, NT = 48, CH = 8;
  const size_t tensor_bytes = static_cast<size_t>(S) * T * I * 4;
  float* host; float* dev;
  CK(hipHostMalloc(reinterpret_cast<void**>(&host), tensor_bytes * NT, hipHostMallocDefault));
  CK(hipMalloc(reinterpret_cast<void**>(&dev), tensor_bytes * NT));
  for (size_t i = 0; i < tensor_bytes * NT / 4; i += 1024) host[i] = static_cast<float>(i);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gb = tensor_bytes * NT / 1e9;
  auto report = [&](const char* name, double host_ms) {
    float ms; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f ms  %6.1f GB/s  (host issue %.1f ms)\n", name, ms, gb / (ms * 1e-3), host_ms);
  };
  for (int rep = 0; rep < 2; ++rep) {
    auto h0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < NT; ++k)
      CK(hipMemcpyAsync(dev + k * (tensor_bytes / 4), host + k * (tensor_bytes / 4), tensor_bytes, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(e1, st));
    double hm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    report("A contiguous hipMemcpyAsync per tensor", hm);
  }
  for (int ch : {8, 16, 4}) {
    auto h0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, st));
    for (int t0 = 0; t0 < T; t0 += ch)
      for (int k = 0; k < NT; ++k)
        CK(hipMemcpy2DAsync(dev + k * (tensor_bytes / 4) + static_cast<size_t>(t0) * I, static_cast<size_t>(T) * I * 4,
                            host + k * (tensor_bytes / 4) + static_cast<size_t>(t0) * I, static_cast<size_t>(T) * I * 4,
                            static_cast<size_t>(ch) * I * 4, S, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(e1, st));
    double hm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    char nm[96]; snprintf(nm, sizeof nm, "B hipMemcpy2DAsync, chunk %d steps", ch);
    report(nm, hm);
  }
  std::vector<Desc> hd(NT);
  for (int k = 0; k < NT; ++k) hd[k] = Desc{host + k * (tensor_bytes / 4), dev + k * (tensor_bytes / 4), S};
  Desc* dd; CK(hipMalloc(reinterpret_cast<void**>(&dd), sizeof(Desc) * NT));
  CK(hipMemcpy(dd, hd.data(), sizeof(Desc) * NT, hipMemcpyHostToDevice));
  for (int blocks : {4, 16, 64}) {
    auto h0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, st));
    for (int t0 = 0; t0 < T; t0 += CH)
      hipLaunchKernelGGL(pull_chunk, dim3(blocks, NT), dim3(256), 0, st, dd, NT, T, I, t0, t0 + CH);
    CK(hipEventRecord(e1, st));
    double hm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    char nm[96]; snprintf(nm, sizeof nm, "C pull kernel, %d x %d workgroups per chunk", blocks, NT);
    report(nm, hm);
  }
  CK(hipGetLastError());
  // check one value
  float v; CK(hipMemcpy(&v, dev + 1024 * 7, 4, hipMemcpyDeviceToHost));
  printf("check %g (want %g)\n", v, 1024.0 * 7);
  return 0;
}
