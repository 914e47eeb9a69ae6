// Development tool: what does this MI355X sustain on bare fp32 MFMA issue (no LDS, no memory)?
// Prints TFLOP/s and the in-kernel shader clock (s_memtime / s_memrealtime) for 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_peak(float* out, unsigned long long* stamps, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD
    const int blocks = ncu * wps, threads = 256;      // 4 waves per block = 1 per SIMD
    float* out; unsigned long long* st;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&st, (size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(threads), 0, 0, out, st, 1000, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(threads), 0, 0, out, st, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 /*mfma per iter*/ * (32.0 * 32 * 2 * 2);
    printf("waves/SIMD %d: %.1f TFLOP/s  in-kernel clock median %.0f MHz (p10 %.0f p90 %.0f)  kernel %.2f ms\n", wps,
           flops / (ms * 1e-3) / 1e12, clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], ms);
    hipFree(out); hipFree(st);
  }
  return 0;
}
